// Fused FeedForward of the C = 320 level, second form (round 5): v_mfma_f32_32x32x16, ONE wave per SIMD that owns 400+
// registers, every tensor of the operator in registers in the TRANSPOSED orientation, the GEGLU issued between the matrix
// instructions of the next hidden chunk.
//
//   out = alpha * ( GEGLU( LN?(X) . W1^T + b1 ) . W2^T + b2 + rowvec[m / rows_per_vec] ) + beta1 * res1 + beta2 * res2
//
// (FeedForward with GEGLU: dp/models/attention.py:1185-1243, activations.py:93-123; optionally the LayerNorm in front of it,
// attention.py:540-567 / 745-756.)  Same operator and C ABI contract as wiw_ffn_geglu_bf16 (ffn.hip), different weight tiling.
//
// Why a second form.  ffn.hip (round 3) splits a CU's eight waves into two roles so that the GEGLU of one wave runs beside the
// MFMAs of its SIMD partner; round 5 measured what that buys (tools/ubench/mfma_fill.hip): a VALU stream beside a partner's
// MFMA stream gets ONE issue per MFMA, while a wave's OWN independent VALU work hides completely behind its own 32x32x16 MFMAs
// up to four instructions per MFMA.  The exact-erf GELU is 2.8 VALU instructions per 32x32x16 MFMA of this operator: it fits.
// ffn.hip keeps the matrix pipe 0.36 busy (0.81 PFLOP/s); this form issues 60 MFMAs per 32 hidden units with four VALU
// instructions in each gap of the up-projection and none in the down-projection.
//
// Work item = 128 rows (tokens) of X per block, 32 per wave.  Transposed orientation (as csrc/attention32.hip):
//   phase 1 (hidden chunk c = 32 units):  V^T, G^T [32 units x 32 tokens] = W1c[32 x 320] . X^T        A = W1 rows (LDS), B = X^T (80 VGPRs, whole tile)
//   GEGLU:  H^T = (V^T + b_v) * gelu_erf(G^T + b_g)                                                    in the accumulator registers
//   phase 2:  Y^T [320 x 32 tokens] += W2c[320 x 32] . H^T                                             A = W2 rows (LDS), B = H^T (8 VGPRs)
// * C/D layout of 32x32x16: lane (token = lane & 31, hi = lane >> 5), register r <-> row (r & 3) + 8 (r >> 2) + 4 hi.  W1 row
//   key(i) = 16 (i >> 4) + 8 ((i >> 2) & 1) + 4 ((i >> 3) & 1) + (i & 3) feeds MFMA row i, so registers 8 ks .. 8 ks + 7 of a lane
//   are the EIGHT CONSECUTIVE hidden units 16 ks + 8 hi .. + 7: the packed H is the B operand of phase 2 as it stands.
// * Pipeline: iteration i issues  S1: phase 1 of chunk i (40 MFMAs, V and G accumulators alternating) with the GEGLU of chunk
//   i-1 in its gaps (20 stages of 4 packed / 4 plain VALU instructions per half block = 4 per gap), then
//   S2: phase 2 of chunk i-1 (20 MFMAs over ten Y accumulators).  V / G accumulators are double-buffered (64 VGPRs), Y is 160.
// * Weights stream through LDS by LDS-DMA, one contiguous KiB per instruction (hip.TiledW(sw16=True): 8 rows x 128 B blocks whose
//   16-byte chunks sit at slot c ^ ((row >> 1) & 7) — every 16-lane group of a ds_read_b128 covers all banks once): W1 chunks
//   (40 KiB) double-buffered, ONE W2 tile (320 x 64 units = 40 KiB, two chunks) refilled while phase 1 runs, b1 resident.
//   One barrier per iteration + one in every odd iteration (the W2 refill).  All LDS reads are asm statements with counted
//   waits (hipcc neither sinks them next to their users nor drains the queue at every use).
// * GELU: exact erf by Abramowitz-Stegun 7.1.26 as in common.h (|err| <= 1.5e-7), with the constants folded:
//   h = (0.5 v + 0.5 b_v) * (|g| erf|g| + g); the packed bias carries 0.5 b_v (unet.py pack for this kernel).
// * Epilogue in registers: v_permlane32_swap pairs give every lane 8 consecutive output channels of its token; bias, per-frame
//   vector, residuals (16-bit or fp32) and the store (16-bit or fp32) are 16- / 32-byte accesses.  No LDS staging.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// LDS-DMA, 16 bytes per lane, in the SADDR form: wave-uniform 64-bit block address in SGPRs + one 32-bit lane offset — a piece
// costs no VGPR of its own (the builtin takes a per-lane 64-bit pointer; hipcc hoisted ten of them out of the chunk loop and
// spilled them).  M0 carries the LDS address and is saved / restored inside the statement (cdna guide 5.7); s_nop 4: an SGPR
// operand written just before by VALU (readfirstlane) must not be read by the VMEM instruction too early.
WIW_DEV void glds16s(uint64_t sb, unsigned voff, uint32_t lds) {    // both wave-uniform by construction (SALU values)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds) : "memory");
}
WIW_DEV void glds16(const char* sbase, unsigned voff, char* l) {
#ifdef FF_DMA_BUILTIN
    __builtin_amdgcn_global_load_lds((gptr_t)(sbase + voff), (lptr_t)l, 16, 0, 0);
    return;
#endif
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)l);
    // (the builtin returns a SIGNED int: without the casts a low half with bit 31 set sign-extends over the high half)
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)sbase >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)sbase);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(lds) : "memory");
}

#ifdef WIW_F16
#define FF_MFMA32(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wiw_h16x8, a), __builtin_bit_cast(wiw_h16x8, b), c, 0, 0, 0)
#else
#define FF_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

constexpr int C = 320, HID = 1280, CH = 32, NCH = HID / CH;   // 40 hidden chunks of 32 units
constexpr int KS = C / 16;                                     // 20 k-steps of 16 over X's channels
constexpr int BM = 128;
constexpr int W2_OFF = 0, W2_BYTES = C * 128;                  // one W2 tile: 320 rows x 64 units = 40 KiB
constexpr int W1_OFF = W2_BYTES, W1_SLOT = 2 * CH * C * 2;     // one W1 chunk: [32 value | 32 gate] rows x 320 = 40 KiB, as 5 K tiles of 8 KiB
constexpr int B1_OFF = W1_OFF + 2 * W1_SLOT, B1_BYTES = 2 * HID * 4;
constexpr int SMEM = B1_OFF + B1_BYTES;                        // 133 120 of the 163 840 bytes of a CU

#ifndef FF_ABLATE
#define FF_ABLATE 0   // timing experiments only (results wrong): 1 no GEGLU VALU, 2 no weight DMA in the loop, 4 no LDS fragment reads, 8 no barriers
#endif
#if FF_ABLATE & 4
#define FF_DSR(dst, addr, off) asm volatile("" : "+v"(dst) : "v"(addr))
#else
#define FF_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#endif
#define FF_WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
// a wait that also tells hipcc the named registers are defined HERE (cdna guide 5.7 form ii); only registers whose reads the
// count covers may be named
#define FF_WAIT2(n, a, b) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b))
#define FF_WAIT1(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a))
#define FF_GAP                                              \
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x402, 16, 0);     \
    __builtin_amdgcn_sched_barrier(0);

struct Ffn32Args {
    const void* X;          // [M][ldx] 16-bit: LayerNorm output, or its raw input when ln != 0
    const char* W1;         // TiledW(sw16) of the packed [2560][320] (chunks of [32 value | 32 gate] rows)
    const float* b1;        // packed [2560]: per chunk [32 x 0.5 b_value | 32 x b_gate]
    const char* W2;         // TiledW(sw16) of [320][1280]
    const float* b2;        // [320] or null
    const float* rowvec;    // [M / rows_per_vec][rowvec_ld] or null
    const void* res1;       // [M][ldr1] or null (16-bit, or fp32 with f32 bit 1)
    const void* res2;       // [M][ldr2] or null (bit 2)
    void* out;              // [M][ldo] (fp32 with bit 0)
    int M, ldx, ldo, ldr1, ldr2, rowvec_ld, rows_per_vec, ln, f32;
    float alpha, beta1, beta2, ln_eps;
};

struct P0 { static constexpr int value = 0; };
struct P1 { static constexpr int value = 1; };

__global__ __launch_bounds__(256, 1) void ffn32_kernel(const Ffn32Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int ntiles = (p.M + BM - 1) / BM;
    const int nb = gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;

    // ---- LDS-DMA sources (this lane's 16 bytes of every 1-KiB block)
    // (uniform block address + a 32-bit lane offset: the saddr form of global_load_lds — one VGPR for every piece, not a 64-bit
    // pointer per piece: hipcc hoisted ten of those out of the loop and spilled them)
    const unsigned lane16 = lane * 16;
    auto dma_w1 = [&](int cw, int slot) {   // chunk cw (40 contiguous KiB of the tiled matrix) -> [kt][row][128 B] of the slot
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int b = wave + 4 * k;          // block b = 5 * (row block of 8) + kt
            glds16(p.W1 + ((int64_t)cw * 40 + b) * 1024, lane16, smem + W1_OFF + slot * W1_SLOT + (b % 5) * 8192 + (b / 5) * 1024);
        }
    };
    // ---- LDS read addresses of this lane (A-operand fragments: row of a 32-row block, 16-byte chunk 2 (ks & 3) + hi)
    const int keyi = 16 * (li >> 4) + 8 * ((li >> 2) & 1) + 4 * ((li >> 3) & 1) + (li & 3);
    const int aV = keyi * 128 + ((hi ^ ((keyi >> 1) & 7)) << 4);
    const int aW = li * 128 + ((hi ^ ((li >> 1) & 7)) << 4);
    int w1a[2][4], w2a[4];     // [slot][ks & 3]; gate rows: + 4096; K tile kt: + 8192 kt.  w2a[2 (c & 1) + ks]; out block ob: + 4096 ob
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        w1a[0][q] = W1_OFF + (aV ^ (q << 5));
        w1a[1][q] = W1_OFF + W1_SLOT + (aV ^ (q << 5));
        w2a[q] = W2_OFF + (aW ^ (q << 5));
    }
    const int b1a = B1_OFF + hi * 32;   // bias of units 8 hi .. + 7 of a chunk; + 64: units 16 + 8 hi ..; + 128: the gate half; chunk c: + 256 c

    // ---- state
    bf16x8 xf[KS];             // X^T B operands of the tile: token li, channels 16 ks + 8 hi .. + 7
    f32x16 Y[10];              // Y^T accumulators: out block ob, lane (token, hi), register r <-> channel 32 ob + (r & 3) + 8 (r >> 2) + 4 hi
    f32x16 vgV[2], vgG[2];     // V^T / G^T accumulators of chunk parity 0 / 1
    union { uint32_t u[4]; bf16x8 v; } hb[2];   // packed H^T of the chunk whose phase 2 runs next: key steps 0 / 1
    f32x4 bvq[4], bgq[4];      // bias of the chunk whose GEGLU runs next, in accumulator register order (register r = [r >> 2][r & 3])
    bf16x8 fV[3], fG[3], fW[4];
    f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;

    // GEGLU of one chunk = 68 UNITS of two value pairs each (hipcc unpacks v_pk_*_f32 in the shadow of an MFMA — a packed f32 op
    // beside MFMAs costs as much as its two halves —, so a unit is ~4 VALU issues): half block h (registers 8 h .. 8 h + 7 of
    // V / G = key step h), row k of 17, pairs 2 sub, 2 sub + 1.  Unit u of a chunk: h = u / 34, k = (u % 34) / 2, sub = u & 1.
    // The 68 units are spread over the 60 MFMA gaps of an iteration (4.5 issues per gap; a 32x32x16 MFMA hides 4).
    float gq[8], gax[8], gt[8], ge[8], gp[8];     // live values of the half block in flight
    // (plain v_fma_f32 / v_mul_f32 with LITERAL constants: the packed forms would need every constant as a 64-bit register pair —
    // two dozen registers this kernel does not have — and hipcc splits them beside MFMAs anyway)
    auto geglu_unit = [&](auto par_tag, auto u_tag) {
        constexpr int PAR = decltype(par_tag)::value, U = decltype(u_tag)::value;
        constexpr int H = U / 34, K = (U % 34) / 2, E0 = 4 * (U & 1);
        f32x16& Vv = vgV[PAR];
        f32x16& Gg = vgG[PAR];
#pragma unroll
        for (int e = E0; e < E0 + 4; ++e) {
            const int r = 8 * H + e;     // accumulator register of this value
            if constexpr (K == 0) gq[e] = Gg[r] + bgq[r >> 2][r & 3];
            else if constexpr (K == 1) gax[e] = __builtin_fabsf(gq[e]);
            else if constexpr (K == 2) gt[e] = __builtin_fmaf(0.3275911f * 0.70710678118654752f, gax[e], 1.0f);
            else if constexpr (K == 3) ge[e] = gax[e] * (-1.4426950408889634f * 0.5f);
            else if constexpr (K == 4) ge[e] = ge[e] * gax[e];
            else if constexpr (K == 5) gt[e] = __builtin_amdgcn_rcpf(gt[e]);
            else if constexpr (K == 6) ge[e] = __builtin_amdgcn_exp2f(ge[e]);
            else if constexpr (K == 7) gp[e] = __builtin_fmaf(1.061405429f, gt[e], -1.453152027f);
            else if constexpr (K == 8) gp[e] = __builtin_fmaf(gp[e], gt[e], 1.421413741f);
            else if constexpr (K == 9) gp[e] = __builtin_fmaf(gp[e], gt[e], -0.284496736f);
            else if constexpr (K == 10) gp[e] = __builtin_fmaf(gp[e], gt[e], 0.254829592f);
            else if constexpr (K == 11) gp[e] = -gp[e] * gt[e];
            else if constexpr (K == 12) gp[e] = __builtin_fmaf(gp[e], ge[e], 1.0f);          // erf|g|
            else if constexpr (K == 13) gq[e] = __builtin_fmaf(gax[e], gp[e], gq[e]);        // |g| erf|g| + g = 2 gelu(g)
            else if constexpr (K == 14) gt[e] = __builtin_fmaf(Vv[r], 0.5f, bvq[r >> 2][r & 3]);   // 0.5 (v + b_v)
            else if constexpr (K == 15) gt[e] = gt[e] * gq[e];
        }
        if constexpr (K == 16) {
            hb[H].u[E0 / 2] = pack2bf(gt[E0], gt[E0 + 1]);
            hb[H].u[E0 / 2 + 1] = pack2bf(gt[E0 + 2], gt[E0 + 3]);
        }
    };
    // the units of MFMA gap g of an iteration: half block 0 in gaps 0..24, half block 1 in gaps 25..49 (phase 2 needs H of key
    // step 0 from gap 40 on and of key step 1 from gap 50 on) — 1.36 units = 5.4 VALU issues per gap
    auto geglu_gap = [&](auto par_tag, auto g_tag) {
        constexpr int g = decltype(g_tag)::value;
        if constexpr (g < 50) {
            constexpr int h = g / 25, gg = g % 25;
            constexpr int u0 = 34 * h + gg * 34 / 25, u1 = 34 * h + (gg + 1) * 34 / 25;
            if constexpr (u1 > u0) geglu_unit(par_tag, std::integral_constant<int, u0>{});
            if constexpr (u1 > u0 + 1) geglu_unit(par_tag, std::integral_constant<int, u0 + 1>{});
        }
    };

    // One iteration i of a tile (41 per tile):
    //   S1 (HAS1): phase 1 of chunk i — 20 k-steps, V and G accumulators of parity PAR = i & 1 alternating (40 gaps)
    //   S2 (HAS2): phase 2 of chunk i - 1 — 20 MFMAs over the ten Y accumulators (20 gaps)
    //   GEGLU (HASG = HAS2) of chunk i - 1 in gaps 0..49.
    // LDS-DMA pieces ride in the gaps too: W1 chunk `next_cw` (10 per wave, slots 0..9 of S1), the W2 tile `w2_tile` (slots 10..19;
    // unit tile t serves phase 2 of chunks 2 t, 2 t + 1 = iterations 2 t + 1, 2 t + 2 and is fetched in iteration 2 t + 1, which
    // therefore has a second barrier between S1 and S2).  ONE other barrier, after gap 57: everybody is through
    // phase 1 of chunk i (its W1 slot may be refilled from the next iteration's first gap on), W1(i + 1) has landed (my pieces:
    // vmcnt(0)), my last W2 fragment reads have returned — then the first fragment reads of the next iteration's phase 1.
    auto iteration = [&](auto par_tag, auto has1_tag, auto hasg_tag, auto has2_tag, auto ns_tag, int i, int next_cw, int w2_tile, bool next_has1) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool HAS1 = decltype(has1_tag)::value != 0, HASG = decltype(hasg_tag)::value != 0, HAS2 = decltype(has2_tag)::value != 0;
        constexpr int SLOT = PAR;                 // W1 chunk i sits in slot i & 1
        constexpr int NS = decltype(ns_tag)::value;   // slot of the chunk whose phase 1 runs next (chunk 0 of the next tile: 0)
        using Qt = std::integral_constant<int, 1 - PAR>;   // chunk i - 1: parity of its accumulators; its half of the W2 tile
        // wave-uniform bases of this iteration's DMA pieces, made opaque: hipcc otherwise computes every piece's 64-bit address
        // for every unrolled iteration up front and spills 200 SGPRs into VGPR lanes (1 800 v_readlane / v_writelane)
        uint64_t w1n = (uint64_t)(uintptr_t)p.W1 + (uint64_t)((next_cw >= 0 ? next_cw : 0) * 40) * 1024;
        uint64_t w2n = (uint64_t)(uintptr_t)p.W2 + (uint64_t)(w2_tile >= 0 ? w2_tile : 0) * 1024;
        uint32_t w1dst = (uint32_t)(uintptr_t)(lptr_t)(smem + W1_OFF) + (next_cw & 1) * W1_SLOT;
        asm volatile("" : "+s"(w1n), "+s"(w2n), "+s"(w1dst));
        auto dma_piece = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (FF_ABLATE & 2) return;
            if (s < 10 && next_cw >= 0) {
                const int b = wave + 4 * s;
                glds16s(w1n + (uint64_t)b * 1024, lane16, w1dst + (b % 5) * 8192 + (b / 5) * 1024);
            }
            if (s >= 10 && w2_tile >= 0) {
                const int nbk = wave + 4 * (s - 10);
                glds16s(w2n + (uint64_t)nbk * (HID / 64) * 1024, lane16, (uint32_t)(uintptr_t)(lptr_t)(smem + W2_OFF) + nbk * 1024);
            }
        };
        const bool refill = w2_tile >= 0;      // (wave-uniform, the same for every wave of the block)
        // ---- S1
        if constexpr (HAS1) {
            auto slot = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                if constexpr (s + 2 < KS) {
                    constexpr int s2 = s + 2;
                    FF_DSR(fV[s2 % 3], w1a[SLOT][s2 & 3], (s2 >> 2) * 8192);
                    FF_DSR(fG[s2 % 3], w1a[SLOT][s2 & 3], (s2 >> 2) * 8192 + 4096);
                }
                dma_piece(s_tag);
                if constexpr (s + 2 < KS) FF_WAIT2(4, fV[s % 3], fG[s % 3]);
                else if constexpr (s == 18) FF_WAIT2(2, fV[s % 3], fG[s % 3]);
                else FF_WAIT2(0, fV[s % 3], fG[s % 3]);
                if constexpr (s == 0) vgV[PAR] = FF_MFMA32(fV[0], xf[0], zero);
                else vgV[PAR] = FF_MFMA32(fV[s % 3], xf[s], vgV[PAR]);
                if constexpr (HASG && !(FF_ABLATE & 1)) geglu_gap(Qt{}, std::integral_constant<int, 2 * s>{});
                FF_GAP
                if constexpr (s == 0) vgG[PAR] = FF_MFMA32(fG[0], xf[0], zero);
                else vgG[PAR] = FF_MFMA32(fG[s % 3], xf[s], vgG[PAR]);
                if constexpr (HASG && !(FF_ABLATE & 1)) geglu_gap(Qt{}, std::integral_constant<int, 2 * s + 1>{});
                FF_GAP
            };
#define FF_S20(f)                                                                                                                 \
    f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});                \
    f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});                \
    f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{}); f(std::integral_constant<int, 8>{});                \
    f(std::integral_constant<int, 9>{}); f(std::integral_constant<int, 10>{}); f(std::integral_constant<int, 11>{});              \
    f(std::integral_constant<int, 12>{}); f(std::integral_constant<int, 13>{}); f(std::integral_constant<int, 14>{});             \
    f(std::integral_constant<int, 15>{}); f(std::integral_constant<int, 16>{}); f(std::integral_constant<int, 17>{});             \
    f(std::integral_constant<int, 18>{}); f(std::integral_constant<int, 19>{});
            FF_S20(slot)
        } else {   // the last two iterations of a tile: no phase 1 to hide behind
            auto nslot = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                dma_piece(s_tag);
                if constexpr (HASG && !(FF_ABLATE & 1)) {
                    geglu_gap(Qt{}, std::integral_constant<int, 2 * s>{});
                    geglu_gap(Qt{}, std::integral_constant<int, 2 * s + 1>{});
                }
            };
            FF_S20(nslot)
        }
        // ---- an iteration that refilled the W2 tile: visible to everybody before phase 2 reads it
        if (refill) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(FF_ABLATE & 8)) __syncthreads();
        }
        // ---- S2
        {
            if constexpr (HAS2) {
                FF_DSR(fW[0], w2a[2 * (1 - PAR) + 0], 0);
                FF_DSR(fW[1], w2a[2 * (1 - PAR) + 0], 4096);
                FF_DSR(fW[2], w2a[2 * (1 - PAR) + 0], 2 * 4096);
            }
            const int ba = b1a + i * 256;    // the bias of chunk i for the GEGLU of the next iteration: fetched in gap 50 (this
                                             // iteration's GEGLU reads its bias until gap ~47)
            auto p2 = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                constexpr int ks = j / 10, ob = j % 10;
                (void)&fV; (void)&fG; (void)&w1a; (void)&bvq; (void)&bgq; (void)next_has1; (void)ba;   // (named here: clang does not capture what only a discarded `if constexpr` branch uses)
                if constexpr (j == 10 && HAS1) {
                    FF_DSR(bvq[0], ba, 0);   FF_DSR(bvq[1], ba, 16);
                    FF_DSR(bvq[2], ba, 64);  FF_DSR(bvq[3], ba, 80);
                    FF_DSR(bgq[0], ba, 128); FF_DSR(bgq[1], ba, 144);
                    FF_DSR(bgq[2], ba, 192); FF_DSR(bgq[3], ba, 208);
                }
                if constexpr (HAS2) {
                    if constexpr (j + 3 < 20) {
                        constexpr int j3 = j + 3;
                        FF_DSR(fW[j3 % 4], w2a[2 * (1 - PAR) + j3 / 10], (j3 % 10) * 4096);
                    }
                    if constexpr (j >= 10 && j < 13 && HAS1) FF_WAIT1(11, fW[j % 4]);   // (the eight bias reads sit between fW[j] and fW[j + 3])
                    else if constexpr (j < 17) FF_WAIT1(3, fW[j % 4]);
                    else if constexpr (j == 17) FF_WAIT1(2, fW[j % 4]);
                    else FF_WAIT1(4, fW[j % 4]);      // (everything older than the four reads issued behind the barrier has landed)
                    Y[ob] = FF_MFMA32(fW[j % 4], hb[ks].v, Y[ob]);
                }
                if constexpr (HASG && !(FF_ABLATE & 1)) geglu_gap(Qt{}, std::integral_constant<int, 40 + j>{});
                FF_GAP
                if constexpr (j == 17) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if constexpr (HAS1) asm volatile("" : "+v"(bvq[0]), "+v"(bvq[1]), "+v"(bvq[2]), "+v"(bvq[3]), "+v"(bgq[0]), "+v"(bgq[1]), "+v"(bgq[2]), "+v"(bgq[3]));
                    if (!(FF_ABLATE & 8)) __syncthreads();
                    if (next_has1) {     // first fragment reads of the next iteration's phase 1
                        FF_DSR(fV[0], w1a[NS][0], 0);
                        FF_DSR(fG[0], w1a[NS][0], 4096);
                        FF_DSR(fV[1], w1a[NS][1], 0);
                        FF_DSR(fG[1], w1a[NS][1], 4096);
                    } else {             // keep the queue depth the waits of gaps 58 / 59 count on
                        FF_DSR(fV[0], w1a[0][0], 0); FF_DSR(fG[0], w1a[0][0], 0); FF_DSR(fV[1], w1a[0][0], 0); FF_DSR(fG[1], w1a[0][0], 0);
                    }
                }
            };
            FF_S20(p2)
#undef FF_S20
        }
    };

    // ---- kernel prologue: the resident bias, the first W1 chunk
    for (int k = wave; k < B1_BYTES / 1024; k += 4) glds16((const char*)p.b1 + k * 1024, lane16, smem + B1_OFF + k * 1024);
    dma_w1(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FF_DSR(fV[0], w1a[0][0], 0);
    FF_DSR(fG[0], w1a[0][0], 4096);
    FF_DSR(fV[1], w1a[0][1], 0);
    FF_DSR(fG[1], w1a[0][1], 4096);
    {   // the bias of chunk 0
        FF_DSR(bvq[0], b1a, 0);   FF_DSR(bvq[1], b1a, 16);
        FF_DSR(bvq[2], b1a, 64);  FF_DSR(bvq[3], b1a, 80);
        FF_DSR(bgq[0], b1a, 128); FF_DSR(bgq[1], b1a, 144);
        FF_DSR(bgq[2], b1a, 192); FF_DSR(bgq[3], b1a, 208);
        FF_WAIT(0);
        asm volatile("" : "+v"(bvq[0]), "+v"(bvq[1]), "+v"(bvq[2]), "+v"(bvq[3]), "+v"(bgq[0]), "+v"(bgq[1]), "+v"(bgq[2]), "+v"(bgq[3]),
                     "+v"(fV[0]), "+v"(fG[0]), "+v"(fV[1]), "+v"(fG[1]));
    }

    const int ntl = (ntiles - (int)blockIdx.x + nb - 1) / nb;
    for (int ti = 0; ti < ntl; ++ti) {
        const int tile = blockIdx.x + ti * nb;
        const int m = tile * BM + wave * 32 + li;
        const int mc = m < p.M ? m : p.M - 1;           // rows past M are computed on a copy of the last row and never stored
        // ---- X^T operands of the tile (+ the LayerNorm without its affine: gamma is folded into W1, beta into b1)
        {
            const uint16_t* src = (const uint16_t*)p.X + (int64_t)mc * p.ldx + hi * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(src + ks * 16);
            if (p.ln) {
                float s = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xf[ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const wiw_f32x2 lh = unpack2(x.u[j]); s += lh.x + lh.y; }
                }
                const float mean = xor32_sum(s) * (1.0f / (float)C);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xf[ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const wiw_f32x2 lh = unpack2(x.u[j]);
                        const float a = lh.x - mean, b = lh.y - mean;
                        q = __builtin_fmaf(a, a, __builtin_fmaf(b, b, q));
                    }
                }
                const float rstd = rsqrtf(xor32_sum(q) * (1.0f / (float)C) + p.ln_eps);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xf[ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const wiw_f32x2 lh = unpack2(x.u[j]);
                        x.u[j] = pack2bf((lh.x - mean) * rstd, (lh.y - mean) * rstd);
                    }
                    xf[ks] = x.v;
                }
            }
        }
#pragma unroll
        for (int ob = 0; ob < 10; ++ob) Y[ob] = zero;

        const bool more = ti + 1 < ntl;
        // iteration i: phase 1 of chunk i, GEGLU and phase 2 of chunk i - 1.  W1 chunk i + 1 is fetched during S1 of iteration i;
        // unit tile t of W2 (chunks 2 t, 2 t + 1) during S1 of iteration 2 t + 1, where its first reader runs.
        iteration(P0{}, P1{}, P0{}, P0{}, P1{}, 0, 1, -1, true);
        for (int i = 1; i + 1 < NCH; i += 2) {
            iteration(P1{}, P1{}, P1{}, P1{}, P0{}, i, i + 1, (i - 1) >> 1, true);
            iteration(P0{}, P1{}, P1{}, P1{}, P1{}, i + 1, i + 2, -1, true);
        }
        iteration(P1{}, P1{}, P1{}, P1{}, P0{}, NCH - 1, -1, (NCH - 2) >> 1, false);
        iteration(P0{}, P0{}, P1{}, P1{}, P0{}, NCH, more ? 0 : -1, -1, more);

        // ---- epilogue: lane (token, hi) holds channels 32 ob + 8 g + 4 hi + (0..3) in registers 4 g .. 4 g + 3; a swap of the even /
        // odd groups between the two lanes of a token gives the lower lane channels 16 jj .. + 7, the upper 16 jj + 8 .. + 15
        {
            const bool ok = m < p.M;
            const float al = p.alpha;
            const bool out_f32 = (p.f32 & 1) != 0, r1_f32 = (p.f32 & 2) != 0, r2_f32 = (p.f32 & 4) != 0;
            const float* rv = p.rowvec ? p.rowvec + (int64_t)(mc / p.rows_per_vec) * p.rowvec_ld : nullptr;
            auto ld8 = [&](const void* base, bool f32, int64_t off, float (&v)[8]) {
                if (f32) {
                    const float4 a = *(const float4*)((const float*)base + off), b = *(const float4*)((const float*)base + off + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
                    unpack8(*(const uint4*)((const uint16_t*)base + off), v);
                }
            };
#pragma unroll
            for (int ob = 0; ob < 10; ++ob)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Y[ob][8 * jj + e]), __float_as_uint(Y[ob][8 * jj + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw[0]);
                        v[4 + e] = __uint_as_float(sw[1]);
                    }
                    const int n = 32 * ob + 16 * jj + 8 * hi;
                    if (p.b2) {
                        const float4 a = *(const float4*)(p.b2 + n), b = *(const float4*)(p.b2 + n + 4);
                        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                    }
                    if (rv) {
                        const float4 a = *(const float4*)(rv + n), b = *(const float4*)(rv + n + 4);
                        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= al;
                    if (p.res1) {
                        float f[8];
                        ld8(p.res1, r1_f32, (int64_t)mc * p.ldr1 + n, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += p.beta1 * f[e];
                    }
                    if (p.res2) {
                        float f[8];
                        ld8(p.res2, r2_f32, (int64_t)mc * p.ldr2 + n, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += p.beta2 * f[e];
                    }
                    if (ok) {
                        if (out_f32) {
                            float* d = (float*)p.out + (int64_t)m * p.ldo + n;
                            *(float4*)d = float4{v[0], v[1], v[2], v[3]};
                            *(float4*)(d + 4) = float4{v[4], v[5], v[6], v[7]};
                        } else {
                            *(uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + n) = pack8(v);
                        }
                    }
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int wiw_ffn32_geglu(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                               const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1,
                               int ldr1, float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out,
                               int ldo, int64_t M, int C_in, int hidden, int ln, float ln_eps, int f32) {
    WIW_REQUIRE(X && W1 && b1 && W2 && out, "ffn32_geglu: null X / W1 / b1 / W2 / out pointer");
    WIW_REQUIRE(C_in == C && hidden == HID, "ffn32_geglu: built for C = 320, hidden = 1280 (the UNet's first level); use wiw_gemm_bf16 elsewhere");
    WIW_REQUIRE(M > 0 && M < (1ll << 31) - BM, "ffn32_geglu: bad M");
    WIW_REQUIRE(ldx % 8 == 0 && ldx >= C && ldo % 8 == 0 && ldo >= C, "ffn32_geglu: ldx / ldo must be multiples of 8 and >= 320");
    WIW_REQUIRE(res1 == nullptr || (ldr1 % 8 == 0 && ldr1 >= C), "ffn32_geglu: ldr1 must be a multiple of 8 and >= 320");
    WIW_REQUIRE(res2 == nullptr || (ldr2 % 8 == 0 && ldr2 >= C), "ffn32_geglu: ldr2 must be a multiple of 8 and >= 320");
    WIW_REQUIRE(rowvec == nullptr || (rows_per_vec > 0 && rowvec_ld % 4 == 0 && rowvec_ld >= C), "ffn32_geglu: bad rowvec_ld / rows_per_vec");
    WIW_REQUIRE((((uintptr_t)X | (uintptr_t)W1 | (uintptr_t)b1 | (uintptr_t)W2 | (uintptr_t)b2 | (uintptr_t)rowvec | (uintptr_t)res1 |
                  (uintptr_t)res2 | (uintptr_t)out) & 15) == 0, "ffn32_geglu: pointers must be 16-byte aligned");
    WIW_REQUIRE(!ln || ln_eps > 0.0f, "ffn32_geglu: the fused LayerNorm needs ln_eps > 0");
    WIW_REQUIRE(f32 >= 0 && f32 < 8, "ffn32_geglu: f32 is a 3-bit mask (out, res1, res2)");
    static std::once_flag once;
    static bool attr_ok = false;
    static int num_cu = 256;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute((const void*)ffn32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
    });
    if (!attr_ok) {
        wiw_set_error("hipFuncSetAttribute(ffn32) failed");
        return WIW_ELAUNCH;
    }
    Ffn32Args a;
    a.X = X; a.W1 = (const char*)W1; a.b1 = b1; a.W2 = (const char*)W2; a.b2 = b2;
    a.rowvec = rowvec; a.res1 = res1; a.res2 = res2; a.out = out;
    a.M = (int)M; a.ldx = ldx; a.ldo = ldo; a.ldr1 = ldr1; a.ldr2 = ldr2; a.rowvec_ld = rowvec_ld;
    a.rows_per_vec = rows_per_vec > 0 ? rows_per_vec : 1; a.ln = ln; a.f32 = f32;
    a.alpha = alpha; a.beta1 = beta1; a.beta2 = beta2; a.ln_eps = ln_eps;
    const int tiles = (int)((M + BM - 1) / BM);
    const int grid = tiles < num_cu ? tiles : num_cu;
    hipLaunchKernelGGL(ffn32_kernel, dim3((unsigned)grid), dim3(256), SMEM, (hipStream_t)stream, a);
    return wiw_check_launch("wiw_ffn32_geglu");
}
