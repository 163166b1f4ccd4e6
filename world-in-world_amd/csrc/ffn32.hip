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
WIW_DEV void glds16(const char* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

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
    const char* w1src = p.W1 + lane * 16;
    const char* w2src = p.W2 + lane * 16;
    auto dma_w1 = [&](int cw, int slot) {   // chunk cw (40 contiguous KiB of the tiled matrix) -> [kt][row][128 B] of the slot
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int b = wave + 4 * k;          // block b = 5 * (row block of 8) + kt
            glds16(w1src + ((int64_t)cw * 40 + b) * 1024, smem + W1_OFF + slot * W1_SLOT + (b % 5) * 8192 + (b / 5) * 1024);
        }
    };
    auto dma_w2 = [&](int t) {              // unit tile t (K tile t of the [320][1280] matrix): row blocks 0..39
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int nbk = wave + 4 * k;
            glds16(w2src + ((int64_t)nbk * (HID / 64) + t) * 1024, smem + W2_OFF + nbk * 1024);
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

    // GEGLU of one half block (registers 8 h .. 8 h + 7 of V / G = key step h), stage st of 20, four instructions each
    f32x2_t gq[4], gax[4], gt[4], ge[4], gp[4];     // live values of the half block in flight
    // FF_PIN4: an empty asm that reads and writes the four pairs of a stage — every pair then lives in an aligned 64-bit
    // register pair (the packed VOP3P forms need that; without it instruction selection falls back to scalar v_fma_f32) and
    // nothing of the next stage is computed before it
#define FF_PIN4(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]))
    auto geglu_stage = [&](auto par_tag, auto half_tag, auto st_tag) {
        constexpr int PAR = decltype(par_tag)::value, H = decltype(half_tag)::value, ST = decltype(st_tag)::value;
        f32x16& Vv = vgV[PAR];
        f32x16& Gg = vgG[PAR];
        const f32x2_t one = {1.0f, 1.0f};
        if constexpr (ST == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gq[e] = f32x2_t{Gg[8 * H + 2 * e], Gg[8 * H + 2 * e + 1]} + f32x2_t{bgq[(8 * H + 2 * e) >> 2][(2 * e) & 3], bgq[(8 * H + 2 * e) >> 2][(2 * e + 1) & 3]};
            FF_PIN4(gq);
        } else if constexpr (ST == 1 || ST == 2) {
            constexpr int o = (ST - 1) * 2;
#pragma unroll
            for (int e = o; e < o + 2; ++e) gax[e] = f32x2_t{__builtin_fabsf(gq[e].x), __builtin_fabsf(gq[e].y)};
        } else if constexpr (ST == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gt[e] = __builtin_elementwise_fma(f32x2_t{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, gax[e], one);
            FF_PIN4(gt);
        } else if constexpr (ST == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ge[e] = gax[e] * (-1.4426950408889634f * 0.5f);
            FF_PIN4(ge);
        } else if constexpr (ST == 5) {
#pragma unroll
            for (int e = 0; e < 4; ++e) ge[e] = ge[e] * gax[e];
            FF_PIN4(ge);
        } else if constexpr (ST == 6 || ST == 7) {
            constexpr int o = (ST - 6) * 2;
#pragma unroll
            for (int e = o; e < o + 2; ++e) gt[e] = f32x2_t{__builtin_amdgcn_rcpf(gt[e].x), __builtin_amdgcn_rcpf(gt[e].y)};
        } else if constexpr (ST == 8 || ST == 9) {
            constexpr int o = (ST - 8) * 2;
#pragma unroll
            for (int e = o; e < o + 2; ++e) ge[e] = f32x2_t{__builtin_amdgcn_exp2f(ge[e].x), __builtin_amdgcn_exp2f(ge[e].y)};
        } else if constexpr (ST == 10) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(f32x2_t{1.061405429f, 1.061405429f}, gt[e], f32x2_t{-1.453152027f, -1.453152027f});
            FF_PIN4(gp);
        } else if constexpr (ST == 11) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], f32x2_t{1.421413741f, 1.421413741f});
            FF_PIN4(gp);
        } else if constexpr (ST == 12) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], f32x2_t{-0.284496736f, -0.284496736f});
            FF_PIN4(gp);
        } else if constexpr (ST == 13) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], f32x2_t{0.254829592f, 0.254829592f});
            FF_PIN4(gp);
        } else if constexpr (ST == 14) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = -gp[e] * gt[e];
            FF_PIN4(gp);
        } else if constexpr (ST == 15) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], ge[e], one);          // erf|g|
            FF_PIN4(gp);
        } else if constexpr (ST == 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gq[e] = __builtin_elementwise_fma(gax[e], gp[e], gq[e]);       // |g| erf|g| + g = 2 gelu(g)
            FF_PIN4(gq);
        } else if constexpr (ST == 17) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                gt[e] = __builtin_elementwise_fma(f32x2_t{Vv[8 * H + 2 * e], Vv[8 * H + 2 * e + 1]}, f32x2_t{0.5f, 0.5f}, f32x2_t{bvq[(8 * H + 2 * e) >> 2][(2 * e) & 3], bvq[(8 * H + 2 * e) >> 2][(2 * e + 1) & 3]});
            FF_PIN4(gt);
        } else if constexpr (ST == 18) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gt[e] = gt[e] * gq[e];
            FF_PIN4(gt);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) hb[H].u[e] = pack2bf(gt[e].x, gt[e].y);
        }
    };

    // One iteration: S1 = phase 1 of chunk i (parity PAR = i & 1) with the GEGLU of chunk i - 1 in its gaps; S2 = phase 2 of
    // chunk i - 1.  HAS1 / HAS2: the first iteration of a tile has no S2 / GEGLU, the 41st no S1.
    auto iteration = [&](auto par_tag, auto has1_tag, auto has2_tag, int i, int next_cw) {   // next_cw: W1 chunk to fetch now (-1: none)
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool HAS1 = decltype(has1_tag)::value != 0, HAS2 = decltype(has2_tag)::value != 0;
        constexpr int SLOT = PAR;                 // W1 chunk i sits in slot i & 1
        constexpr int Q = 1 - PAR;                // chunk i - 1: parity of its accumulators, and its half of the W2 tile
        // ---- top: my DMA pieces of W1(i) have landed, my LDS reads of the previous iteration are done; everybody is through
        // S2(i-1): the other W1 slot and (odd i) the W2 tile may be refilled
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!(FF_ABLATE & 8)) __syncthreads();
        if (next_cw >= 0 && !(FF_ABLATE & 2)) dma_w1(next_cw, next_cw & 1);   // (its slot was last read by phase 1 of chunk i - 1)
        if (PAR == 1 && !(FF_ABLATE & 2)) dma_w2((i - 1) >> 1);               // unit tile t serves phase 2 of chunks 2 t, 2 t + 1 = S2 of iterations 2 t + 1, 2 t + 2
        if (HAS1) {
            FF_DSR(fV[0], w1a[SLOT][0], 0);
            FF_DSR(fG[0], w1a[SLOT][0], 4096);
            FF_DSR(fV[1], w1a[SLOT][1], 0);
            FF_DSR(fG[1], w1a[SLOT][1], 4096);
        }
        // ---- S1
        if constexpr (HAS1) {
            auto slot = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                constexpr int kt = s >> 2;
                if constexpr (s + 2 < KS) {
                    constexpr int s2 = s + 2;
                    FF_DSR(fV[s2 % 3], w1a[SLOT][s2 & 3], (s2 >> 2) * 8192);
                    FF_DSR(fG[s2 % 3], w1a[SLOT][s2 & 3], (s2 >> 2) * 8192 + 4096);
                }
                if constexpr (PAR == 0 && HAS2 && s == 18) {   // phase-2 prefetch (even iterations: the W2 tile is already there)
                    FF_DSR(fW[0], w2a[2 * Q + 0], 0);
                    FF_DSR(fW[1], w2a[2 * Q + 0], 4096);
                }
                if constexpr (PAR == 0 && HAS2 && s == 19) FF_DSR(fW[2], w2a[2 * Q + 0], 2 * 4096);
                (void)kt;
                if constexpr (s + 2 < KS || (PAR == 0 && HAS2 && s == 18)) FF_WAIT2(4, fV[s % 3], fG[s % 3]);
                else if constexpr (PAR == 0 && HAS2 && s == 19) FF_WAIT2(3, fV[s % 3], fG[s % 3]);
                else if constexpr (s == 18) FF_WAIT2(2, fV[s % 3], fG[s % 3]);
                else FF_WAIT2(0, fV[s % 3], fG[s % 3]);
                if constexpr (s == 0) vgV[PAR] = FF_MFMA32(fV[0], xf[0], zero);
                else vgV[PAR] = FF_MFMA32(fV[s % 3], xf[s], vgV[PAR]);
                if constexpr (HAS2 && !(FF_ABLATE & 1)) geglu_stage(std::integral_constant<int, Q>{}, std::integral_constant<int, (s >= 10)>{}, std::integral_constant<int, 2 * (s % 10)>{});
                FF_GAP
                if constexpr (s == 0) vgG[PAR] = FF_MFMA32(fG[0], xf[0], zero);
                else vgG[PAR] = FF_MFMA32(fG[s % 3], xf[s], vgG[PAR]);
                if constexpr (HAS2 && !(FF_ABLATE & 1)) geglu_stage(std::integral_constant<int, Q>{}, std::integral_constant<int, (s >= 10)>{}, std::integral_constant<int, 2 * (s % 10) + 1>{});
                FF_GAP
            };
            slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{}); slot(std::integral_constant<int, 2>{});
            slot(std::integral_constant<int, 3>{}); slot(std::integral_constant<int, 4>{}); slot(std::integral_constant<int, 5>{});
            slot(std::integral_constant<int, 6>{}); slot(std::integral_constant<int, 7>{}); slot(std::integral_constant<int, 8>{});
            slot(std::integral_constant<int, 9>{}); slot(std::integral_constant<int, 10>{}); slot(std::integral_constant<int, 11>{});
            slot(std::integral_constant<int, 12>{}); slot(std::integral_constant<int, 13>{}); slot(std::integral_constant<int, 14>{});
            slot(std::integral_constant<int, 15>{}); slot(std::integral_constant<int, 16>{}); slot(std::integral_constant<int, 17>{});
            slot(std::integral_constant<int, 18>{}); slot(std::integral_constant<int, 19>{});
        } else if constexpr (HAS2) {   // the 41st iteration: the GEGLU of the last chunk without MFMAs to hide behind
            auto st = [&](auto h_tag, auto s_tag) { geglu_stage(std::integral_constant<int, Q>{}, h_tag, s_tag); };
#define FF_ALLST(h)                                                                                                                  \
    st(h, std::integral_constant<int, 0>{}); st(h, std::integral_constant<int, 1>{}); st(h, std::integral_constant<int, 2>{});       \
    st(h, std::integral_constant<int, 3>{}); st(h, std::integral_constant<int, 4>{}); st(h, std::integral_constant<int, 5>{});       \
    st(h, std::integral_constant<int, 6>{}); st(h, std::integral_constant<int, 7>{}); st(h, std::integral_constant<int, 8>{});       \
    st(h, std::integral_constant<int, 9>{}); st(h, std::integral_constant<int, 10>{}); st(h, std::integral_constant<int, 11>{});     \
    st(h, std::integral_constant<int, 12>{}); st(h, std::integral_constant<int, 13>{}); st(h, std::integral_constant<int, 14>{});    \
    st(h, std::integral_constant<int, 15>{}); st(h, std::integral_constant<int, 16>{}); st(h, std::integral_constant<int, 17>{});    \
    st(h, std::integral_constant<int, 18>{}); st(h, std::integral_constant<int, 19>{});
            FF_ALLST(P0{})
            FF_ALLST(P1{})
#undef FF_ALLST
        }
        // ---- S2: phase 2 of chunk i - 1 (its half Q of the W2 tile), bias reads for the GEGLU of chunk i
        if constexpr (HAS2) {
            if constexpr (PAR == 1 || !HAS1) {   // odd iterations: the W2 tile was refilled at the top — make it visible first
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(FF_ABLATE & 8)) __syncthreads();
                FF_DSR(fW[0], w2a[2 * Q + 0], 0);
                FF_DSR(fW[1], w2a[2 * Q + 0], 4096);
                FF_DSR(fW[2], w2a[2 * Q + 0], 2 * 4096);
            }
            const int ba = b1a + i * 256;
            if constexpr (HAS1) {   // the bias of chunk i in accumulator register order (4 x 16 B value half, 4 x 16 B gate half)
                FF_DSR(bvq[0], ba, 0);   FF_DSR(bvq[1], ba, 16);
                FF_DSR(bvq[2], ba, 64);  FF_DSR(bvq[3], ba, 80);
                FF_DSR(bgq[0], ba, 128); FF_DSR(bgq[1], ba, 144);
                FF_DSR(bgq[2], ba, 192); FF_DSR(bgq[3], ba, 208);
            }
            auto p2 = [&](auto j_tag) {
                constexpr int j = decltype(j_tag)::value;
                constexpr int ks = j / 10, ob = j % 10;
                if constexpr (j + 3 < 20) {
                    constexpr int j3 = j + 3;
                    FF_DSR(fW[j3 % 4], w2a[2 * Q + j3 / 10], (j3 % 10) * 4096);
                }
                if constexpr (j < 3 && HAS1) FF_WAIT1(11, fW[j % 4]);
                else if constexpr (j < 3) FF_WAIT1(3, fW[j % 4]);
                else if constexpr (j + 3 < 20) FF_WAIT1(3, fW[j % 4]);
                else if constexpr (j == 17) FF_WAIT1(2, fW[j % 4]);
                else if constexpr (j == 18) FF_WAIT1(1, fW[j % 4]);
                else FF_WAIT1(0, fW[j % 4]);
                Y[ob] = FF_MFMA32(fW[j % 4], hb[ks].v, Y[ob]);
                FF_GAP
            };
            p2(std::integral_constant<int, 0>{}); p2(std::integral_constant<int, 1>{}); p2(std::integral_constant<int, 2>{});
            p2(std::integral_constant<int, 3>{}); p2(std::integral_constant<int, 4>{}); p2(std::integral_constant<int, 5>{});
            p2(std::integral_constant<int, 6>{}); p2(std::integral_constant<int, 7>{}); p2(std::integral_constant<int, 8>{});
            p2(std::integral_constant<int, 9>{}); p2(std::integral_constant<int, 10>{}); p2(std::integral_constant<int, 11>{});
            p2(std::integral_constant<int, 12>{}); p2(std::integral_constant<int, 13>{}); p2(std::integral_constant<int, 14>{});
            p2(std::integral_constant<int, 15>{}); p2(std::integral_constant<int, 16>{}); p2(std::integral_constant<int, 17>{});
            p2(std::integral_constant<int, 18>{}); p2(std::integral_constant<int, 19>{});
            if constexpr (HAS1) asm volatile("" : "+v"(bvq[0]), "+v"(bvq[1]), "+v"(bvq[2]), "+v"(bvq[3]), "+v"(bgq[0]), "+v"(bgq[1]), "+v"(bgq[2]), "+v"(bgq[3]));
        } else {   // first iteration of a tile: only the bias of chunk 0 is fetched
            const int ba = b1a;
            FF_DSR(bvq[0], ba, 0);   FF_DSR(bvq[1], ba, 16);
            FF_DSR(bvq[2], ba, 64);  FF_DSR(bvq[3], ba, 80);
            FF_DSR(bgq[0], ba, 128); FF_DSR(bgq[1], ba, 144);
            FF_DSR(bgq[2], ba, 192); FF_DSR(bgq[3], ba, 208);
            FF_WAIT(0);
            asm volatile("" : "+v"(bvq[0]), "+v"(bvq[1]), "+v"(bvq[2]), "+v"(bvq[3]), "+v"(bgq[0]), "+v"(bgq[1]), "+v"(bgq[2]), "+v"(bgq[3]));
        }
    };

    // ---- kernel prologue: the resident bias, the first W1 chunk
    for (int k = wave; k < B1_BYTES / 1024; k += 4) glds16((const char*)p.b1 + k * 1024 + lane * 16, smem + B1_OFF + k * 1024);
    dma_w1(0, 0);

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
        iteration(P0{}, P1{}, P0{}, 0, 1);
        for (int i = 1; i + 1 < NCH; i += 2) {
            iteration(P1{}, P1{}, P1{}, i, i + 1);
            iteration(P0{}, P1{}, P1{}, i + 1, i + 2);
        }
        iteration(P1{}, P1{}, P1{}, NCH - 1, -1);
        iteration(P0{}, P0{}, P1{}, NCH, more ? 0 : -1);      // (chunk 0 of the next tile goes to slot 0, last read by phase 1 of chunk 38)

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
