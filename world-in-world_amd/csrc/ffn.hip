// Fused FeedForward for the C = 320 level of the UNet on gfx950 (MI355X):
//
//   out = alpha * ( GEGLU( LN?(X) . W1^T + b1 ) . W2^T + b2 + rowvec[m / rows_per_vec] ) + beta1 * res1 + beta2 * res2
//
// (FeedForward with GEGLU: dp/models/attention.py:1185-1243, activations.py:93-123; optionally the LayerNorm in front of
// it, attention.py:540-567 / 745-756.)  The [M][1280] hidden tensor — 660 MB written and read back per layer at
// M = 258 048 — never exists, and the K = 320 up-projection, whose 5-K-tile output tiles spent a third of their time in
// an epilogue with the matrix pipe idle, becomes part of ONE long MFMA stream per 128-row tile.
//
// Work item = 128 rows of X.  The hidden dimension is walked in 20 chunks of 64 units; a chunk is
//   phase 1:  Hc[128][64 v | 64 g] = X[128][320] . W1c[128][320]^T          (K = 320: 5 K tiles of 64)
//   GEGLU  :  H[128][64] = (v + b_v) * gelu_erf(g + b_g)  -> 16-bit, to LDS
//   phase 2:  Y[128][320] += H[128][64] . W2c[320][64]^T                      (K = 64)
// 8 waves, TWO ROLES (one wave of each role per SIMD: waves w and w + 4 share a SIMD):
//   * H-waves 0..3: rows 32w..32w+31.  X fragments live in REGISTERS for the whole tile (80 VGPRs), so phase 1 reads only
//     W1 fragments from LDS (8 ds_read_b128 per 16 MFMAs); accumulators 32 rows x 128 columns (64 VGPRs); then the GEGLU
//     on the VALU and 8-byte LDS writes of H;
//   * Y-waves 4..7: own Y[32 rows][320] (160 accumulator VGPRs) for the whole tile, issue EVERY LDS-DMA instruction of the
//     block (W1 K tiles, W2 chunk, bias chunk) while the H-wave of their SIMD streams MFMAs, and run phase 2 of chunk c-1
//     (80 MFMAs) exactly while that H-wave evaluates the GEGLU of chunk c on the VALU — matrix pipe and VALU of a SIMD
//     are busy with different waves instead of taking turns inside one.
// The roles are one chunk apart (H_c goes through a double-buffered LDS tile), H-waves run ahead across tile boundaries
// (next tile's X loads are issued under the last GEGLU), the Y-waves' epilogue (bias, per-frame vector, residuals,
// 16-byte streaming stores through a per-wave LDS transpose) happens once per 629 MFLOP instead of once per 26.
// One block barrier per slot; 6 slots per chunk: 5 K tiles + the GEGLU / phase-2 slot.
//
// Weights: W1 packed [2560][320] in chunks of 128 rows = [64 value | 64 gate] and W2 [320][1280], both in the tiled
// layout of wiw_gemm_bf16 (1-KiB blocks, chunks pre-swizzled): one DMA instruction copies one contiguous KiB and the LDS
// image is the XOR-swizzled row-major tile the fragment reads expect.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"

#ifndef FFN_P2_PRIO
#define FFN_P2_PRIO 0   // s_setprio of the Y-waves' phase-2 MFMA stream (A/B knob)
#endif
#ifndef FFN_GE_PRIO
#define FFN_GE_PRIO 0   // s_setprio of the H-waves' GEGLU segment (A/B knob)
#endif
#ifndef FFN_W1_BULK
#define FFN_W1_BULK 1   // 1 (round 6): the Y-waves issue ALL five W1 K tiles of the next chunk in slot 5 (in front of their phase-2
                        // MFMAs, while the H-waves evaluate the GEGLU) instead of one tile per K-tile slot: 648-654 -> 628-634 us at
                        // M = 258 048 (profiles/r19f_ffn_dma_schedules.txt); 0 = the schedule of rounds 3-5 (A/B knob)
#endif
#ifndef FFN_FREE_K
#define FFN_FREE_K 1    // (needs FFN_W1_BULK) 1 (round 6): with a chunk's five W1 K tiles resident before its first K tile is read, the
                        // H-waves run the K phase WITHOUT the per-K-tile block barriers: two barriers per chunk (A: W1 resident /
                        // W2 buffer and ring hand-over; B: W2 + bias landed, H of the previous chunk visible) instead of six:
                        // 634-646 -> 615-620 us at M = 258 048 (profiles/r19g_ffn_freek.txt; timelines r19h); 0 = A/B knob
#endif
static_assert(!FFN_FREE_K || FFN_W1_BULK, "FFN_FREE_K needs the bulk W1 schedule");
#ifndef FFN_BIAS_INIT
#define FFN_BIAS_INIT FFN_FREE_K   // (needs FFN_FREE_K) 1 (round 6): the up-projection's accumulators START at the bias (b1's chunk is
                                   // resident before the chunk's K phase: the Y-waves issue it with the bulk W1 tiles) instead of at
                                   // zero — the GEGLU loses its 32 bias adds per lane and chunk (of ~300 VALU issues); 0 = A/B knob
#endif
static_assert(!FFN_BIAS_INIT || FFN_FREE_K, "FFN_BIAS_INIT needs the free-running K phase (bias resident at barrier A)");
#ifndef FFN_ABLATE
#define FFN_ABLATE 0   // timing-only builds (tools/build_variant.py): 1 no GELU math, 2 no phase-2 MFMAs, 3 no LDS-DMA, 4 no phase-1 MFMAs
#endif

namespace {

constexpr int C = 320, HID = 1280, HC = 64, NCH = HID / HC;      // 20 hidden chunks
constexpr int BM = 128, BK = 64, NKT = C / BK;                   // 5 K tiles per chunk
constexpr int KS = C / 32;                                       // 10 k-steps of 32 over X's K
#ifndef FFN_RING
#define FFN_RING 5
#endif
constexpr int RING_STAGES = FFN_RING;
[[maybe_unused]] constexpr int LEAD = RING_STAGES - 1;     // W1 K tiles: issued LEAD slots before they are read
constexpr int RING_STAGE_BYTES = 2 * HC * BK * 2;                 // one W1 K tile: 128 rows x 64 k = 16 KiB
constexpr int W2_OFF = RING_STAGES * RING_STAGE_BYTES;           // 81920
constexpr int W2_BYTES = C * HC * 2;                             // 320 rows x 64 k = 40 KiB
constexpr int H_OFF = W2_OFF + W2_BYTES;                         // 122880
constexpr int H_BYTES = BM * HC * 2;                             // 16 KiB per parity
constexpr int BIAS_OFF = H_OFF + 2 * H_BYTES;                    // 155648
constexpr int BIAS_BYTES = 2 * HC * 4;                           // 512 B per parity
constexpr int SMEM = BIAS_OFF + 2 * BIAS_BYTES;                  // 156672 of the 163840 bytes of a CU
#ifdef WIW_FFN_TRACE
constexpr int SMEM_LAUNCH = SMEM + 1024;                         // + the trace stamps
#else
constexpr int SMEM_LAUNCH = SMEM;
#endif
constexpr int STG_ROWB = 336;                                    // epilogue staging: 160 columns + 16 B skew
constexpr int STG_WAVE = 16 * STG_ROWB;                          // 5376 B per Y-wave, inside the W2 buffer
static_assert(4 * STG_WAVE <= W2_BYTES, "epilogue staging must fit in the W2 buffer");

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA (16 bytes per lane, LDS destination = wave-uniform base + 16 * lane) as INLINE ASM: hipcc models the builtin as a
// FLAT access that may touch LDS, and while one is pending every compiler-inserted wait for a ds_read is lgkmcnt(0) — the
// Y-waves' software-pipelined fragment reads of phase 2 (8 in flight) would drain at every use.  The asm form is invisible to
// that bookkeeping: its completion is counted by hand (wait_vmcnt below), exactly as for the builtin.  M0 carries the LDS
// address and is saved / restored inside the statement (the compiler does not preserve it around asm).
WIW_DEV void glds16(const char* g, char* l) {
#if FFN_ABLATE == 3
    asm volatile("" ::"v"(g), "s"(l));
    return;
#endif
    const uint32_t lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lptr_t)l);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// X rows and residual rows are read exactly once by the whole launch: streaming (nt) loads, so that they do not evict the
// 2.4 MB of weights every CU of the XCD re-reads from its L2 for every tile (A/B knob FFN_PLAIN_LOADS)
template <typename T>
WIW_DEV T ld_stream(const T* p) {
#ifdef FFN_PLAIN_LOADS
    return *p;
#else
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    static_assert(sizeof(T) == 16, "16-byte loads");
    return __builtin_bit_cast(T, __builtin_nontemporal_load((const i32x4_t*)p));
#endif
}
template <int N>
WIW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
WIW_DEV void wait_vmcnt_rt(int n) {   // wave-uniform n (s_waitcnt takes an immediate)
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 11: wait_vmcnt<11>(); break;
        case 12: wait_vmcnt<12>(); break;
        case 13: wait_vmcnt<13>(); break;
        case 14: wait_vmcnt<14>(); break;
        case 15: wait_vmcnt<15>(); break;
        case 16: wait_vmcnt<16>(); break;
        case 17: wait_vmcnt<17>(); break;
        case 18: wait_vmcnt<18>(); break;
        case 19: wait_vmcnt<19>(); break;
        case 20: wait_vmcnt<20>(); break;
        case 21: wait_vmcnt<21>(); break;
        case 22: wait_vmcnt<22>(); break;
        case 23: wait_vmcnt<23>(); break;
        case 24: wait_vmcnt<24>(); break;
        case 25: wait_vmcnt<25>(); break;
        case 26: wait_vmcnt<26>(); break;
        case 27: wait_vmcnt<27>(); break;
        case 28: wait_vmcnt<28>(); break;
        case 29: wait_vmcnt<29>(); break;
        case 30: wait_vmcnt<30>(); break;
        default: wait_vmcnt<31>(); break;
    }
}
WIW_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
WIW_DEV void slot_barrier() {   // this wave's LDS accesses are retired, then rendezvous of all 8 waves
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef WIW_FFN_TRACE   // debug build: block 0 stamps s_memtime at the slot boundaries of two steady-state chunks (waves 0 and 4)
__device__ long long g_ftrace[2][64];
__device__ unsigned g_fhwid[8];
constexpr int TR_CC0 = 23, TR_NCC = 2;   // chunks 3 and 4 of the block's second tile
// (stamps go to LDS and are copied out at the end: a global store per stamp sits in the vmcnt queue the Y-waves' counted
// waits watch, and made every slot look ~500 cycles longer)
#define FTP(role, idx)                                                                                           \
    do {                                                                                                         \
        if (trace_blk && cc >= TR_CC0 && cc < TR_CC0 + TR_NCC)                                                   \
            *(volatile long long*)(smem + SMEM + ((role) * 64 + (cc - TR_CC0) * 32 + (idx)) * 8) = __builtin_readcyclecounter(); \
    } while (0)
#else
#define FTP(role, idx) do { } while (0)
#endif

// gelu_erf_f of common.h (exact-erf GELU, Abramowitz-Stegun 7.1.26, same arithmetic bit for bit) on EIGHT values as four
// PACKED pairs, breadth first.  Why (tools/ubench/valu_rate.hip, tools/ffn_trace.py): one wave issues at most one VALU
// instruction per ~5.4 cycles (8.5 when it depends on the previous one, ~9 for v_exp / v_rcp) whether it is packed or not
// and whether or not a second wave shares the SIMD — so the cost of the GEGLU is its INSTRUCTION COUNT times 5.4 if every
// stage has independent work, and v_pk_* halves the count.  Evaluated value by value (a loop over gelu_erf_f) the 32
// values of a wave took 3.8 k cycles, dependent-chain-bound.
//   REGP8: an empty asm statement that reads and writes the eight pairs of a stage — every pair exists in its own registers
//   there and nothing of the next stage can be computed before it (source order and sched_barrier do not survive
//   instruction selection for pure arithmetic).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define REGP8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
WIW_DEV f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
WIW_DEV void gelu_erf16(float (&x)[16]) {   // 8 packed pairs per stage: a stage's issue time (~45 cycles) covers a packed result's latency
#if !WIW_GELU_AS7126
    {   // round 5: the sigmoid form of common.h's gelu_erf_f, the same arithmetic bit for bit (13 instead of 18 issues per pair)
        f32x2_t xv[8], a[8], p[8];
        const auto bc = [](float c) { return f32x2_t{c, c}; };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xv[i] = f32x2_t{x[2 * i], x[2 * i + 1]};
            a[i] = f32x2_t{__builtin_fminf(fabsf(x[2 * i]), WIW_GELU_CLAMP), __builtin_fminf(fabsf(x[2 * i + 1]), WIW_GELU_CLAMP)};
            p[i] = pk_fma(bc(WIW_GELU_C4), a[i], bc(WIW_GELU_C3));
        }
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = pk_fma(p[i], a[i], bc(WIW_GELU_C2));
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = pk_fma(p[i], a[i], bc(WIW_GELU_C1));
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = pk_fma(p[i], a[i], bc(WIW_GELU_C0));
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = xv[i] * p[i];
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = f32x2_t{__builtin_amdgcn_exp2f(p[i].x), __builtin_amdgcn_exp2f(p[i].y)};
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = p[i] + bc(1.0f);
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = f32x2_t{__builtin_amdgcn_rcpf(p[i].x), __builtin_amdgcn_rcpf(p[i].y)};
        REGP8(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = xv[i] * p[i];
        REGP8(xv);
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[2 * i] = xv[i].x; x[2 * i + 1] = xv[i].y; }
        return;
    }
#endif
    f32x2_t xv[8], ax[8], z[8], t[8], e[8], poly[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xv[i] = f32x2_t{x[2 * i], x[2 * i + 1]};
        ax[i] = f32x2_t{fabsf(x[2 * i]), fabsf(x[2 * i + 1])};
        z[i] = ax[i] * 0.70710678118654752f;
    }
    REGP8(z);
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] = pk_fma(f32x2_t{0.3275911f, 0.3275911f}, z[i], f32x2_t{1.0f, 1.0f}); e[i] = (z[i] * -1.4426950408889634f) * z[i]; }
    REGP8(t); REGP8(e);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        t[i] = f32x2_t{__builtin_amdgcn_rcpf(t[i].x), __builtin_amdgcn_rcpf(t[i].y)};
        e[i] = f32x2_t{__builtin_amdgcn_exp2f(e[i].x), __builtin_amdgcn_exp2f(e[i].y)};
    }
    REGP8(t); REGP8(e);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = pk_fma(f32x2_t{1.061405429f, 1.061405429f}, t[i], f32x2_t{-1.453152027f, -1.453152027f});
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = pk_fma(poly[i], t[i], f32x2_t{1.421413741f, 1.421413741f});
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = pk_fma(poly[i], t[i], f32x2_t{-0.284496736f, -0.284496736f});
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = pk_fma(poly[i], t[i], f32x2_t{0.254829592f, 0.254829592f});
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = -poly[i] * t[i];
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) poly[i] = pk_fma(poly[i], e[i], f32x2_t{1.0f, 1.0f});     // erf|x|
    REGP8(poly);
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = pk_fma(ax[i], poly[i], xv[i]) * 0.5f;
    REGP8(xv);
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[2 * i] = xv[i].x; x[2 * i + 1] = xv[i].y; }
}

struct FfnArgs {
    const uint16_t* X;      // [M][ldx]   LayerNorm output, or its raw input when ln != 0
    const uint16_t* W1;     // packed + tiled [2560][320]
    const float* b1;        // packed [2560]
    const uint16_t* W2;     // tiled [320][1280]
    const float* b2;        // [320] or null
    const float* rowvec;    // [M / rows_per_vec][rowvec_ld] or null
    const uint16_t* res1;   // [M][ldr1] or null
    const uint16_t* res2;   // [M][ldr2] or null
    uint16_t* out;          // [M][ldo]
    int M, ldx, ldo, ldr1, ldr2, rowvec_ld, rows_per_vec, ln;
    int f32;                // F32E: bit 0 out, bit 1 res1, bit 2 res2 are fp32 (leading dimensions in ELEMENTS)
    float alpha, beta1, beta2, ln_eps;
    uint16_t* out16;        // F32E (ABI 16): a second, 16-bit copy of the output rows [M][ldo16], or null
    int ldo16;
};

// F32E (round 4, the fp32 residual stream of ABI 11 in this kernel): res1 / res2 / out may be fp32 (p.f32 bits 1 / 2 / 0); the
// Y accumulators then take bias, per-frame vector and residuals in the FRAGMENT layout (a lane holds 4 consecutive columns of
// a row: 16-byte fp32 / 8-byte 16-bit accesses) and are rounded once, or not at all — no 16-bit staging in between.
// F32X (round 6, ABI 16: f32 bit 3): X itself is the fp32 residual stream and the LayerNorm (ln != 0) reads it — the H-waves
// stage a row block's 80 fp32 values per lane in registers, take exact two-pass moments and pack the normalised 16-bit MFMA
// operand; the standalone LayerNorm pass of the fp32-stream mode (read 4 B + write 2 B per element) is gone.
template <bool F32E, bool F32X = false>
__global__ __launch_bounds__(512, 2) void ffn_kernel(const FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int wq = wave & 3;                            // row quarter of the tile served by this wave (both roles)
    const int ntiles = (p.M + BM - 1) / BM;
    const int nb = gridDim.x;
    if ((int)blockIdx.x >= ntiles) return;
    const int ntl = (ntiles - (int)blockIdx.x + nb - 1) / nb;   // tiles of this block: blockIdx.x + i * nb
    const int NC = ntl * NCH;                                   // chunks of this block; H-waves at chunk cc, Y-waves at cc - 1
#ifdef WIW_FFN_TRACE
    const bool trace_blk = blockIdx.x == 0 && lane == 0 && (wave & 3) == 0;
    if (blockIdx.x == 0 && lane == 0) g_fhwid[wave] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID[15:0]
#endif

    if (wave < 4) {
        // =====================================================================================================
        // H-waves: phase 1 + GEGLU
        // =====================================================================================================
        bf16x8 xa[2][KS];      // X fragments of the tile: rows 16*mi + frow, k = 32*ks + 8*fq .. +7
        f32x4 acc[2][8];       // lane: row 16*mi + frow, chunk columns 16*ni + 4*fq + r  (ni 0..3 value, 4..7 gate)
        bf16x8 fbA[8], fbB[8]; // W1 fragments of k-step 0 / 1 of the K tile in flight
        auto load_x = [&](int tile) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                int m = tile * BM + wq * 32 + mi * 16 + frow;
                m = m < p.M ? m : p.M - 1;   // rows past M are computed on a copy of the last row and never stored
                const uint16_t* src = p.X + (int64_t)m * p.ldx + fq * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xa[mi][ks] = ld_stream((const bf16x8*)(src + ks * 32));
            }
        };
        // LayerNorm without its affine (gamma is folded into W1, beta into b1): two passes over the 320 values of a row,
        // which sit in the 4 lanes (fq) of the row x 10 fragments x 8 values; result rounded to the 16-bit operand type
        auto normalize_x = [&]() {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float s = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xa[mi][ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const wiw_f32x2 lh = unpack2(x.u[j]); s += lh.x + lh.y; }
                }
                const float mean = xor32_sum(xor16_sum(s)) * (1.0f / (float)C);
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xa[mi][ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const wiw_f32x2 lh = unpack2(x.u[j]);
                        const float a = lh.x - mean, b = lh.y - mean;
                        q = __builtin_fmaf(a, a, __builtin_fmaf(b, b, q));
                    }
                }
                const float rstd = rsqrtf(xor32_sum(xor16_sum(q)) * (1.0f / (float)C) + p.ln_eps);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    union { bf16x8 v; uint32_t u[4]; } x;
                    x.v = xa[mi][ks];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const wiw_f32x2 lh = unpack2(x.u[j]);
                        x.u[j] = pack2bf((lh.x - mean) * rstd, (lh.y - mean) * rstd);
                    }
                    xa[mi][ks] = x.v;
                }
            }
        };
        // F32X: row block mi of a tile as fp32 (lane: row 16*mi + frow, columns 32*ks + 8*fq .. +7 = two 16-byte loads), staged in
        // xf0 / xf1; block 0 of the NEXT tile is requested before the last GEGLU of a tile (80 registers in flight under it),
        // block 1 right behind that GEGLU (the accumulators are dead by then), so its latency hides behind block 0's moments
        float4 xf0[KS][2], xf1[KS][2];
        auto load_xf = [&](float4 (&xf)[KS][2], int tile, int mi) {
            int m = tile * BM + wq * 32 + mi * 16 + frow;
            m = m < p.M ? m : p.M - 1;
            const float* src = (const float*)p.X + (int64_t)m * p.ldx + fq * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xf[ks][0] = ld_stream((const float4*)(src + ks * 32));
                xf[ks][1] = ld_stream((const float4*)(src + ks * 32 + 4));
            }
        };
        auto normalize_xf = [&](const float4 (&xf)[KS][2], int mi) {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) s += (xf[ks][h].x + xf[ks][h].y) + (xf[ks][h].z + xf[ks][h].w);
            const float mean = xor32_sum(xor16_sum(s)) * (1.0f / (float)C);
            float q = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = xf[ks][h].x - mean, b = xf[ks][h].y - mean, c = xf[ks][h].z - mean, d = xf[ks][h].w - mean;
                    q = __builtin_fmaf(a, a, __builtin_fmaf(b, b, __builtin_fmaf(c, c, __builtin_fmaf(d, d, q))));
                }
            const float rstd = rsqrtf(xor32_sum(xor16_sum(q)) * (1.0f / (float)C) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                union { bf16x8 v; uint32_t u[4]; } x;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    x.u[2 * h] = pack2bf((xf[ks][h].x - mean) * rstd, (xf[ks][h].y - mean) * rstd);
                    x.u[2 * h + 1] = pack2bf((xf[ks][h].z - mean) * rstd, (xf[ks][h].w - mean) * rstd);
                }
                xa[mi][ks] = x.v;
            }
        };
        const int sw0 = (fq ^ (frow & 7)) << 4, sw1 = ((4 + fq) ^ (frow & 7)) << 4;   // swizzled chunk of k-step 0 / 1
        auto read_w1 = [&](bf16x8* fb, int stage, int sw) {
            const char* s = smem + stage * RING_STAGE_BYTES + frow * 128 + sw;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) fb[ni] = *(const bf16x8*)(s + ni * 2048);
        };
        auto mma1 = [&](const bf16x8& f, const bf16x8& x0, const bf16x8& x1, int ni) {   // one W1 fragment x both row blocks
#if FFN_ABLATE == 4
            asm volatile("" ::"v"(f), "v"(x0), "v"(x1));
            return;
#endif
            acc[0][ni] = WIW_MFMA(f, x0, acc[0][ni]);
            acc[1][ni] = WIW_MFMA(f, x1, acc[1][ni]);
        };
        auto mma = [&](const bf16x8* fb, const bf16x8& x0, const bf16x8& x1) {
#if FFN_ABLATE == 4
            asm volatile("" ::"v"(fb[0]), "v"(fb[7]), "v"(x0), "v"(x1));
            return;
#endif
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                acc[0][ni] = WIW_MFMA(fb[ni], x0, acc[0][ni]);
                acc[1][ni] = WIW_MFMA(fb[ni], x1, acc[1][ni]);
            }
            __builtin_amdgcn_s_setprio(0);
        };

        if constexpr (F32X) { load_xf(xf0, blockIdx.x, 0); load_xf(xf1, blockIdx.x, 1); }
        else load_x(blockIdx.x);
        int st = 0;          // ring stage of the next K tile
        int c = 0, ti = 0;   // chunk inside the tile, tile counter of this block
        // one hidden chunk (H-wave side); LAST: compile-time "chunk NCH - 1 of its tile" (F32X only, see the loads below)
        auto chunk = [&](const int cc, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
#if !FFN_BIAS_INIT
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            // ---- slots 0..4: K tile kt.  The MFMAs of a K tile's second k-step run in the NEXT slot, under that slot's
            // first fragment reads, so the matrix pipe is not idle while a slot's first reads are in flight.
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                FTP(0, 2 * kt);          // end of the previous slot's work
                if (!FFN_FREE_K || kt == 0) slot_barrier();
                FTP(0, 2 * kt + 1);      // barrier passed
#if FFN_BIAS_INIT
                if (kt == 0) {   // behind barrier A the chunk's bias is resident: both row blocks start at it (a lane's 4 columns)
                    const char* bs0 = smem + BIAS_OFF + (cc & 1) * BIAS_BYTES + fq * 16;
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni) {
                        const float4 b = *(const float4*)(bs0 + (ni < 4 ? ni * 16 : HC + (ni - 4) * 16) * 4);
                        acc[0][ni] = f32x4{b.x, b.y, b.z, b.w};
                        acc[1][ni] = f32x4{b.x, b.y, b.z, b.w};
                    }
                }
#endif
                // two halves of [2 MFMAs | 1 ds_read_b128] x 8, order pinned: the first half runs the PENDING k-step (fragments
                // fbB of the previous slot) while this K tile's k-step-0 fragments arrive, the second half runs k-step 0 while
                // the k-step-1 fragments arrive — the reads' issue slots sit inside the MFMAs' pipe time (left alone the
                // compiler issues 8 reads, then 16 MFMAs: 64 issue cycles per half with the matrix pipe draining)
                {
                    const char* sp = smem + st * RING_STAGE_BYTES + frow * 128;
                    if (kt > 0) {
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int ni = 0; ni < 8; ++ni) {
                            mma1(fbB[ni], xa[0][2 * kt - 1], xa[1][2 * kt - 1], ni);
                            fbA[ni] = *(const bf16x8*)(sp + sw0 + ni * 2048);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        read_w1(fbA, st, sw0);
                        __builtin_amdgcn_s_setprio(1);
                    }
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni) {
                        mma1(fbA[ni], xa[0][2 * kt], xa[1][2 * kt], ni);
                        fbB[ni] = *(const bf16x8*)(sp + sw1 + ni * 2048);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                st = st + 1 == RING_STAGES ? 0 : st + 1;
            }
            // ---- slot 5: last k-step, then the GEGLU on the VALU while the Y-wave of this SIMD runs phase 2 of chunk cc - 1
            FTP(0, 10);
            slot_barrier();
            FTP(0, 11);
            mma(fbB, xa[0][KS - 1], xa[1][KS - 1]);
            FTP(0, 12);              // last k-step issued
            if constexpr (F32X) {   // X of the next tile: latency under the GEGLU
                // (LAST is compile time and the load unconditional — behind the block's last tile it re-reads row M - 1 — so
                // that the staging registers are provably dead during the other 19 chunks: with a run-time test hipcc kept
                // all 160 of them alive around the loop, 500 bytes of scratch per lane)
                if constexpr (LAST) load_xf(xf0, blockIdx.x + (ti + 1) * nb, 0);
            } else {
                if (c == NCH - 1 && ti + 1 < ntl) load_x(blockIdx.x + (ti + 1) * nb);
            }
            if (FFN_GE_PRIO) __builtin_amdgcn_s_setprio(FFN_GE_PRIO);
            {
                const char* bs = smem + BIAS_OFF + (cc & 1) * BIAS_BYTES;
                char* hb = smem + H_OFF + (cc & 1) * H_BYTES;
#pragma unroll
                for (int np = 0; np < 2; ++np) {      // two 16-column fragments (ni = 2*np, 2*np + 1) per pass: 16 gate values
                    float4 bv[2], bg[2];
                    float g16[16];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ni = 2 * np + h;
#if FFN_BIAS_INIT
                        (void)bs; bv[h] = bg[h] = float4{0.f, 0.f, 0.f, 0.f};      // the accumulators carry the bias already
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                            for (int r = 0; r < 4; ++r) g16[h * 8 + mi * 4 + r] = acc[mi][ni + 4][r];
#else
                        bv[h] = *(const float4*)(bs + (ni * 16 + fq * 4) * 4);
                        bg[h] = *(const float4*)(bs + (HC + ni * 16 + fq * 4) * 4);
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            g16[h * 8 + mi * 4 + 0] = acc[mi][ni + 4][0] + bg[h].x; g16[h * 8 + mi * 4 + 1] = acc[mi][ni + 4][1] + bg[h].y;
                            g16[h * 8 + mi * 4 + 2] = acc[mi][ni + 4][2] + bg[h].z; g16[h * 8 + mi * 4 + 3] = acc[mi][ni + 4][3] + bg[h].w;
                        }
#endif
                    }
#if FFN_ABLATE != 1
                    gelu_erf16(g16);
#endif
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ni = 2 * np + h;
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            const f32x4 v = acc[mi][ni];
                            const float* g = g16 + h * 8 + mi * 4;
                            uint2 pk;
#if FFN_BIAS_INIT
                            pk.x = pack2bf(v[0] * g[0], v[1] * g[1]);
                            pk.y = pack2bf(v[2] * g[2], v[3] * g[3]);
#else
                            pk.x = pack2bf((v[0] + bv[h].x) * g[0], (v[1] + bv[h].y) * g[1]);
                            pk.y = pack2bf((v[2] + bv[h].z) * g[2], (v[3] + bv[h].w) * g[3]);
#endif
                            // H[row][16*ni + 4*fq .. +3]: 16-byte chunk 2*ni + (fq >> 1) of the row, swizzled like every K tile
                            const int row = wq * 32 + mi * 16 + frow;
                            *(uint2*)(hb + row * 128 + (((2 * ni + (fq >> 1)) ^ (frow & 7)) << 4) + (fq & 1) * 8) = pk;
                        }
                    }
                    FTP(0, 20 + np);
                }
            }
            if (FFN_GE_PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (F32X && LAST) load_xf(xf1, blockIdx.x + (ti + 1) * nb, 1);
            FTP(0, 13);              // GEGLU done, H written
            // tile-end barrier: in this iteration the Y-waves finished a tile (its last chunk is cc - 1) and stage its
            // epilogue in the W2 buffer, which every Y-wave must have stopped reading first
            if (c == 0 && cc >= 1) slot_barrier();
        };
        using L0 = std::integral_constant<bool, false>; using L1 = std::integral_constant<bool, true>;
        if constexpr (F32X) {
            int cc = 0;
            for (ti = 0; ti < ntl; ++ti) {
                normalize_xf(xf0, 0); normalize_xf(xf1, 1);
                for (c = 0; c < NCH - 1; ++c, ++cc) chunk(cc, L0{});
                chunk(cc, L1{});
                ++cc;
            }
        } else {
            for (int cc = 0; cc < NC; ++cc) {
                if (c == 0 && p.ln) normalize_x();
                chunk(cc, L0{});
                if (++c == NCH) { c = 0; ++ti; }
            }
        }
        // drain: the Y-waves finish the last chunk (+ the tile-end barrier, see above)
#pragma unroll
        for (int i = 0; i < (FFN_FREE_K ? 3 : NKT + 2); ++i) slot_barrier();
#ifdef WIW_FFN_TRACE
        if (trace_blk) for (int i = 0; i < 64; ++i) g_ftrace[0][i] = *(volatile long long*)(smem + SMEM + i * 8);
#endif
    } else {
        // =====================================================================================================
        // Y-waves: all LDS-DMA of the block, phase 2, tile epilogue
        // =====================================================================================================
        f32x4 accY[2][20];     // lane: row 16*mi + frow, output columns 16*nj + 4*fq + r
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 20; ++nj) accY[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* const W1b = (const char*)p.W1;
        const char* const W2b = (const char*)p.W2;
        // loader state: next W1 K tile to issue = K tile ld_kt of chunk ld_c (of any tile: the weights repeat), stage ld_st
        int ld_c = 0, ld_kt = 0, ld_st = 0;
        int ld_left = NC * NKT;                        // W1 K tiles still to be issued
        auto issue_w1 = [&]() {   // this wave's quarter (4 x 1 KiB = rows 32*wq .. +31 of the chunk) of one W1 K tile
            const char* src = W1b + ((int64_t)((ld_c * 16 + wq * 4) * NKT + ld_kt)) * 1024 + lane * 16;
            char* dst = smem + ld_st * RING_STAGE_BYTES + wq * 4096;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(src + (int64_t)i * NKT * 1024, dst + i * 1024);
            if (++ld_kt == NKT) { ld_kt = 0; ld_c = ld_c + 1 == NCH ? 0 : ld_c + 1; }
            ld_st = ld_st + 1 == RING_STAGES ? 0 : ld_st + 1;
            --ld_left;
        };
#if FFN_W1_BULK
        // prologue: the five W1 K tiles of chunk 0; tile 0 has landed when <= 16 instructions are outstanding
#pragma unroll
        for (int i = 0; i < NKT; ++i) issue_w1();
#if FFN_BIAS_INIT
        // ... and the bias of chunk 0: the H-waves' accumulators start at it (behind barrier A)
        if (lane < 8) glds16((const char*)p.b1 + wq * 128 + lane * 16, smem + BIAS_OFF + wq * 128);
#endif
        wait_vmcnt<FFN_FREE_K ? 0 : 4 * (NKT - 1)>();
        int h1 = 0, h2 = 0;
#else
        // prologue: K tiles 0 .. LEAD-1 (NC >= 20 chunks: they exist); tile 0 has landed when <= 4 * (LEAD - 1) are outstanding
#pragma unroll
        for (int i = 0; i < LEAD; ++i) issue_w1();
        wait_vmcnt<4 * (LEAD - 1)>();
        int h1 = LEAD >= 3 ? 4 : 0, h2 = LEAD >= 4 ? 4 : 0;   // DMA instructions issued in the previous two DMA slots (prologue tiles)
#endif

        const int sw0 = (fq ^ (frow & 7)) << 4, sw1 = ((4 + fq) ^ (frow & 7)) << 4;
        int c = 0;   // chunk (inside its tile) the H-waves work on in this iteration; this wave works on chunk c - 1
        int ti = 0;  // tile counter of the chunk THIS wave works on
        for (int cc = 0; cc <= NC; ++cc) {
            const int c2 = c == 0 ? NCH - 1 : c - 1;   // chunk of phase 2 in this iteration (valid for cc >= 1)
#if FFN_FREE_K
            // ---- K phase of the H-waves (they free-run through the chunk's five resident W1 K tiles): behind barrier A — every
            // Y-wave is through phase 2 of the previous iteration, the W2 buffer is free — this wave issues its share of W2
            // chunk c2 (10 blocks) and of the bias chunk, confirms them, and meets the H-waves at barrier B
            FTP(1, 0);
            slot_barrier();
            FTP(1, 1);
#if !FFN_BIAS_INIT
            if (cc < NC) {
                if (lane < 8) glds16((const char*)p.b1 + c * (2 * HC * 4) + wq * 128 + lane * 16,
                                     smem + BIAS_OFF + (cc & 1) * BIAS_BYTES + wq * 128);
            }
#endif
            if (cc >= 1) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int NB2 = g == 0 ? 4 : 3;
                    const int RB0 = g == 0 ? 0 : 16 + (g - 1) * 12;
                    const int rb = RB0 + wq * NB2;
                    const char* src = W2b + ((int64_t)rb * NCH + c2) * 1024 + lane * 16;
                    char* dst = smem + W2_OFF + rb * 1024;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < NB2) glds16(src + (int64_t)i * NCH * 1024, dst + i * 1024);
                }
            }
            FTP(1, 2);
            wait_vmcnt<0>();
            (void)h1; (void)h2;
#else
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                FTP(1, 3 * kt);          // end of the previous slot's work
                slot_barrier();
                FTP(1, 3 * kt + 1);      // barrier passed
                // The W1 K tile the H-waves read in slot d + 1 was issued LEAD - 1 = 3 DMA slots before slot d: at the end of a
                // slot only what was issued more than two slots ago has to have landed (an LDS-DMA takes ~1.5 k cycles from
                // issue to landed under this kernel's load: with a two-slot lead every slot ended in a 200-350 cycle vmcnt
                // wait, tools/ffn_trace.py).  W2 chunk c2 (40 blocks of 1 KiB, consumed in slot 5) goes out in slots 0..2 as
                // 4 | 3 | 3 blocks per wave (slot 4 therefore waits for everything older than ONE slot), the bias chunk of the
                // H-waves' current chunk (512 B, read by them in slot 5) as 128 B per wave in slot 0.
                int h0 = 0;
                if (kt == 0 && cc < NC) {
                    if (lane < 8) glds16((const char*)p.b1 + c * (2 * HC * 4) + wq * 128 + lane * 16,
                                         smem + BIAS_OFF + (cc & 1) * BIAS_BYTES + wq * 128);
                    h0 += 1;
                }
                if (cc >= 1 && kt < 3) {
                    // (spreading the W2 pieces 3|3|2|2 or 2 x 5 over the slots, or the bulk W1 issue BEHIND phase 2: measured, no
                    // gain — profiles/r19f_ffn_dma_schedules.txt)
                    const int NB2 = kt == 0 ? 4 : 3;              // blocks per wave in this slot (kt is unrolled)
                    const int RB0 = kt == 0 ? 0 : 16 + (kt - 1) * 12;   // first block of this slot: 0 | 16 | 28
                    const int rb = RB0 + wq * NB2;
                    const char* src = W2b + ((int64_t)rb * NCH + c2) * 1024 + lane * 16;
                    char* dst = smem + W2_OFF + rb * 1024;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < NB2) glds16(src + (int64_t)i * NCH * 1024, dst + i * 1024);
                    h0 += NB2;
                }
#if FFN_W1_BULK
                // The chunk's W1 tiles were issued in the previous slot 5 (below), in K-tile order and BEFORE anything issued
                // in this chunk's slots: tile kt + 1 (read by the H-waves in the next slot) has landed once at most the tiles
                // behind it (kt + 2 .. 4) and this chunk's own W2 / bias pieces (h1 accumulates them) are outstanding; W2 and
                // the bias are read in slot 5, so slot 4 waits for everything.
                h1 += h0;
                FTP(1, 3 * kt + 2);  // DMA issued
                wait_vmcnt_rt(kt == 4 ? 0 : 4 * (3 - kt > 0 ? 3 - kt : 0) + h1);
                if (kt == 4) h1 = 0;
                (void)h2;
#else
                if (ld_left > 0) {   // W1 K tile LEAD slots ahead of the one the H-waves read now
                    issue_w1();
                    h0 += 4;
                }
                FTP(1, 3 * kt + 2);  // DMA issued
                wait_vmcnt_rt(h0 + (LEAD >= 3 ? h1 : 0) + (LEAD >= 4 && kt != 4 ? h2 : 0));
                h2 = h1; h1 = h0;
#endif
            }
#endif
            FTP(1, 15);
            slot_barrier();
            FTP(1, 16);
#if FFN_W1_BULK
            // slot 5: every H-wave is past its last read of the ring (slot 4) — the five stages are free.  The next chunk's W1 K
            // tiles go out here, in front of this wave's phase-2 MFMAs: the issue time (the ~110 cycles a VMEM instruction costs
            // its wave, x 20) falls under the H-waves' GEGLU, where this wave otherwise idles ~2 000 cycles, instead of pacing
            // the five K-tile slots (profiles/r05j_ffn_slot_timeline.txt: DMA issue + vmcnt wait 1 100 - 1 500 cycles per slot
            // against 550 - 760 of MFMA work on the H side).  Tile 0 is confirmed at the end of this slot (below).
            const bool bulk = ld_left > 0;
            if (bulk) {
#pragma unroll
                for (int i = 0; i < NKT; ++i) issue_w1();
#if FFN_BIAS_INIT
                // ... and the next chunk's bias (parity buffer last read at the start of chunk cc - 1's K phase): resident,
                // like the W1 tiles, at the end of this iteration — the H-waves' accumulators start at it behind barrier A
                const int cn = c + 1 == NCH ? 0 : c + 1;
                if (lane < 8) glds16((const char*)p.b1 + cn * (2 * HC * 4) + wq * 128 + lane * 16,
                                     smem + BIAS_OFF + ((cc + 1) & 1) * BIAS_BYTES + wq * 128);
#endif
            }
#endif
            if (cc >= 1) {
                // ---- phase 2 of chunk c2: Y += H . W2c^T  (H of the previous iteration, parity (cc - 1) & 1)
                const char* hb = smem + H_OFF + ((cc - 1) & 1) * H_BYTES + (wq * 32 + frow) * 128;
                bf16x8 hf[2][2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    hf[mi][0] = *(const bf16x8*)(hb + mi * 2048 + sw0);
                    hf[mi][1] = *(const bf16x8*)(hb + mi * 2048 + sw1);
                }
                const char* wb = smem + W2_OFF + frow * 128;
                // 40 W2 fragments (k-step ks = j / 20, column block nj = j % 20), each feeding 2 MFMAs, through a ring of 8
                // registers: the order [2 MFMAs | 1 ds_read] is pinned so that 8 reads stay in flight behind the matrix pipe
                // (left alone the compiler issues 2 reads and waits for both before every 4 MFMAs)
                auto wread = [&](int j) { return *(const bf16x8*)(wb + (j % 20) * 2048 + (j >= 20 ? sw1 : sw0)); };
                bf16x8 wf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[j] = wread(j);
                // NO s_setprio here: with this wave at priority 1 the H-wave of the SIMD — evaluating the GEGLU on the VALU at
                // the same time — is starved (its 1.6 k cycles of VALU work ran only AFTER these 80 MFMAs: 4.3 k cycles for the
                // slot, tools/ffn_trace.py); at equal priority the older H-wave keeps its issue slots and the MFMAs, one issue
                // per 16 cycles, take the rest
                __builtin_amdgcn_s_setprio(FFN_P2_PRIO);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 40; ++j) {
#if FFN_ABLATE == 2
                    asm volatile("" ::"v"(wf[j & 7]), "v"(hf[0][j / 20]), "v"(hf[1][j / 20]));
#else
                    accY[0][j % 20] = WIW_MFMA(wf[j & 7], hf[0][j / 20], accY[0][j % 20]);
                    accY[1][j % 20] = WIW_MFMA(wf[j & 7], hf[1][j / 20], accY[1][j % 20]);
#endif
                    if (j + 8 < 40) wf[j & 7] = wread(j + 8);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                FTP(1, 17);          // phase 2 issued
                if (c2 == NCH - 1) {
                    // ---- tile epilogue: 4 passes of 16 rows x 160 columns through a per-wave LDS transpose staged in the
                    // W2 buffer (free once EVERY Y-wave is through phase 2 — the tile-end barrier — until the next slot's
                    // DMA); row-major phase: lane = (row lrow + 3k, 16-byte column chunk lch)
                    slot_barrier();
                    const int tile = blockIdx.x + ti * nb;
                    const int mw0 = tile * BM + wq * 32;
                    char* stg = smem + W2_OFF + wq * STG_WAVE;
                    const int lrow = lane / 20 < 3 ? lane / 20 : 2;   // 60 lanes = 3 rows x 20 chunks; lanes 60..63 repeat row 2
                    const int lch = lane - (lane / 20) * 20;
                    const float al = p.alpha;
                    const bool rv_fast = p.rowvec != nullptr && (p.rows_per_vec % 16) == 0;   // one vector row per 16-row pass
                    auto pass = [&](auto mi_tag, auto half_tag) {
                        constexpr int mi = decltype(mi_tag)::value, half = decltype(half_tag)::value;
                        const int ncol = half * 160 + lch * 8;
                        const int mp0 = mw0 + mi * 16;
                        // every global load of the pass is issued up front (rows past M re-read the last row)
                        float4 b0 = float4{0.f, 0.f, 0.f, 0.f}, b1v = b0, r0 = b0, r1v = b0;
                        uint4 q1[6], q2[6];
                        if (p.b2) { b0 = *(const float4*)(p.b2 + ncol); b1v = *(const float4*)(p.b2 + ncol + 4); }
                        if (rv_fast) {
                            const float* rv = p.rowvec + (int64_t)((mp0 < p.M ? mp0 : p.M - 1) / p.rows_per_vec) * p.rowvec_ld + ncol;
                            r0 = *(const float4*)rv; r1v = *(const float4*)(rv + 4);
                        }
                        if (p.res1) {
#pragma unroll
                            for (int k = 0; k < 6; ++k) {
                                int m = mp0 + (lrow + k * 3 < 16 ? lrow + k * 3 : 15);
                                m = m < p.M ? m : p.M - 1;
                                q1[k] = ld_stream((const uint4*)(p.res1 + (int64_t)m * p.ldr1 + ncol));
                            }
                        }
                        char* wrow = stg + frow * STG_ROWB + fq * 8;
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const f32x4 v = accY[mi][half * 10 + j];
                            uint2 pk;
                            pk.x = pack2bf(v[0] * al, v[1] * al);
                            pk.y = pack2bf(v[2] * al, v[3] * al);
                            *(uint2*)(wrow + j * 32) = pk;
                        }
                        if (p.res2) {   // (after the staging writes: those accumulator registers are free now)
#pragma unroll
                            for (int k = 0; k < 6; ++k) {
                                int m = mp0 + (lrow + k * 3 < 16 ? lrow + k * 3 : 15);
                                m = m < p.M ? m : p.M - 1;
                                q2[k] = ld_stream((const uint4*)(p.res2 + (int64_t)m * p.ldr2 + ncol));
                            }
                        }
                        wave_lds_sync();
                        const float cb[8] = {al * (b0.x + r0.x), al * (b0.y + r0.y), al * (b0.z + r0.z), al * (b0.w + r0.w),
                                             al * (b1v.x + r1v.x), al * (b1v.y + r1v.y), al * (b1v.z + r1v.z), al * (b1v.w + r1v.w)};
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const int rr = lrow + k * 3 < 16 ? lrow + k * 3 : 15;
                            const int m = mp0 + rr;
                            const bool ok = m < p.M;
                            float v[8], f[8];
                            unpack8(*(const uint4*)(stg + rr * STG_ROWB + lch * 16), v);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += cb[e];
                            if (p.rowvec && !rv_fast) {   // odd rows_per_vec (small test shapes): the vector row of every item
                                const float* rv = p.rowvec + (int64_t)((ok ? m : p.M - 1) / p.rows_per_vec) * p.rowvec_ld + ncol;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += al * rv[e];
                            }
                            if (p.res1) {
                                unpack8(q1[k], f);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += p.beta1 * f[e];
                            }
                            if (p.res2) {
                                unpack8(q2[k], f);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += p.beta2 * f[e];
                            }
                            if (ok) {
                                const uint4 ov = pack8(v);
                                uint4* dst = (uint4*)(p.out + (int64_t)m * p.ldo + ncol);
                                __builtin_nontemporal_store(ov.x, &dst->x); __builtin_nontemporal_store(ov.y, &dst->y);
                                __builtin_nontemporal_store(ov.z, &dst->z); __builtin_nontemporal_store(ov.w, &dst->w);
                            }
                        }
                        wave_lds_sync();
                    };
                    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
                    if constexpr (F32E) {
                        const bool out_f32 = (p.f32 & 1) != 0, r1_f32 = (p.f32 & 2) != 0, r2_f32 = (p.f32 & 4) != 0;
                        auto ld4 = [&](const void* base, bool f32, int64_t off) -> float4 {
                            if (f32) return *(const float4*)((const float*)base + off);
                            const uint2 u = *(const uint2*)((const uint16_t*)base + off);
                            const wiw_f32x2 a = unpack2(u.x), b = unpack2(u.y);
                            return float4{a.x, a.y, b.x, b.y};
                        };
#pragma unroll
                        for (int mi = 0; mi < 2; ++mi) {
                            const int m = mw0 + mi * 16 + frow;
                            const bool m_ok = m < p.M;
                            const float* rv = p.rowvec ? p.rowvec + (int64_t)((m_ok ? m : p.M - 1) / p.rows_per_vec) * p.rowvec_ld : nullptr;
#pragma unroll
                            for (int h = 0; h < 4; ++h) {      // quarter passes of 5 column blocks: 2 x 20 residual registers in flight
                                float4 rr1[5], rr2[5];
#pragma unroll
                                for (int j = 0; j < 5; ++j) {
                                    const int n = (h * 5 + j) * 16 + fq * 4;
                                    rr1[j] = (p.res1 && m_ok) ? ld4(p.res1, r1_f32, (int64_t)m * p.ldr1 + n) : float4{0.f, 0.f, 0.f, 0.f};
                                    rr2[j] = (p.res2 && m_ok) ? ld4(p.res2, r2_f32, (int64_t)m * p.ldr2 + n) : float4{0.f, 0.f, 0.f, 0.f};
                                }
#pragma unroll
                                for (int j = 0; j < 5; ++j) {
                                    const int n = (h * 5 + j) * 16 + fq * 4;
                                    if (m_ok) {
                                        f32x4 v = accY[mi][h * 5 + j];
                                        if (p.b2) { const float4 b = *(const float4*)(p.b2 + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                                        if (rv) { const float4 b = *(const float4*)(rv + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                                        v[0] *= al; v[1] *= al; v[2] *= al; v[3] *= al;
                                        const float4 f = rr1[j], g = rr2[j];
                                        v[0] += p.beta1 * f.x + p.beta2 * g.x; v[1] += p.beta1 * f.y + p.beta2 * g.y;
                                        v[2] += p.beta1 * f.z + p.beta2 * g.z; v[3] += p.beta1 * f.w + p.beta2 * g.w;
                                        if (out_f32) {
                                            float* d = (float*)p.out + (int64_t)m * p.ldo + n;
                                            __builtin_nontemporal_store(v[0], d); __builtin_nontemporal_store(v[1], d + 1);
                                            __builtin_nontemporal_store(v[2], d + 2); __builtin_nontemporal_store(v[3], d + 3);
                                        } else {
                                            uint32_t* d = (uint32_t*)(p.out + (int64_t)m * p.ldo + n);
                                            __builtin_nontemporal_store(pack2bf(v[0], v[1]), d);
                                            __builtin_nontemporal_store(pack2bf(v[2], v[3]), d + 1);
                                        }
                                        if (p.out16) {     // the rounded copy a 16-bit consumer (the fused temporal block) reads
                                            uint32_t* d = (uint32_t*)(p.out16 + (int64_t)m * p.ldo16 + n);
                                            __builtin_nontemporal_store(pack2bf(v[0], v[1]), d);
                                            __builtin_nontemporal_store(pack2bf(v[2], v[3]), d + 1);
                                        }
                                    }
                                }
                            }
                        }
                    } else {
                    pass(I0{}, I0{}); pass(I0{}, I1{}); pass(I1{}, I0{}); pass(I1{}, I1{});
                    }
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int nj = 0; nj < 20; ++nj) accY[mi][nj] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ++ti;
                }
            }
#if FFN_W1_BULK
            FTP(1, 18);
            if (bulk) wait_vmcnt<FFN_FREE_K ? 0 : 4 * (NKT - 1)>();   // W1 tile 0 (FFN_FREE_K: every tile) of the next chunk has landed (younger: tiles 1 .. 4; an epilogue's stores make this stricter)
#endif
            c = c + 1 == NCH ? 0 : c + 1;
        }
#ifdef WIW_FFN_TRACE
        if (trace_blk) for (int i = 0; i < 64; ++i) g_ftrace[1][i] = *(volatile long long*)(smem + SMEM + (64 + i) * 8);
#endif
    }
}

}  // namespace

#ifdef WIW_FFN_TRACE
extern "C" int wiw_ffn_trace_read(long long* out) {   // debug builds only: copy the slot-boundary timestamps to the host
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ftrace), sizeof(long long) * 2 * 64) == hipSuccess ? 0 : -1;
}
extern "C" int wiw_ffn_hwid_read(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fhwid), sizeof(unsigned) * 8) == hipSuccess ? 0 : -1;
}
#endif

namespace {
template <bool F32E, bool F32X = false>
int ffn_launch(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
               const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1,
               int ldr1, float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out,
               int ldo, int64_t M, int C_in, int hidden, int ln, float ln_eps, int f32, void* out16 = nullptr, int ldo16 = 0) {
    WIW_REQUIRE(X && W1 && b1 && W2 && out, "ffn_geglu: null X / W1 / b1 / W2 / out pointer");
    WIW_REQUIRE(C_in == C && hidden == HID, "ffn_geglu: built for C = 320, hidden = 1280 (the UNet's first level); use wiw_gemm_bf16 elsewhere");
    WIW_REQUIRE(M > 0 && M < (1ll << 31) - BM, "ffn_geglu: bad M");
    WIW_REQUIRE(ldx % 8 == 0 && ldx >= C && ldo % 8 == 0 && ldo >= C, "ffn_geglu: ldx / ldo must be multiples of 8 and >= 320");
    WIW_REQUIRE(res1 == nullptr || (ldr1 % 8 == 0 && ldr1 >= C), "ffn_geglu: ldr1 must be a multiple of 8 and >= 320");
    WIW_REQUIRE(res2 == nullptr || (ldr2 % 8 == 0 && ldr2 >= C), "ffn_geglu: ldr2 must be a multiple of 8 and >= 320");
    WIW_REQUIRE(rowvec == nullptr || (rows_per_vec > 0 && rowvec_ld % 4 == 0 && rowvec_ld >= C), "ffn_geglu: bad rowvec_ld / rows_per_vec");
    WIW_REQUIRE((((uintptr_t)X | (uintptr_t)W1 | (uintptr_t)b1 | (uintptr_t)W2 | (uintptr_t)b2 | (uintptr_t)rowvec | (uintptr_t)res1 |
                  (uintptr_t)res2 | (uintptr_t)out) & 15) == 0, "ffn_geglu: pointers must be 16-byte aligned");
    WIW_REQUIRE(!ln || ln_eps > 0.0f, "ffn_geglu: the fused LayerNorm needs ln_eps > 0");
    WIW_REQUIRE(out16 == nullptr || (F32E && ldo16 % 8 == 0 && ldo16 >= C && ((uintptr_t)out16 & 15) == 0),
                "ffn_geglu: out16 needs the f32stream entry, ldo16 a multiple of 8 and >= 320, 16-byte alignment");
    WIW_REQUIRE(!F32X || ln, "ffn_geglu: an fp32 X is read by the fused LayerNorm only (ln != 0)");
    static std::once_flag once;
    static bool attr_ok = false;
    static int num_cu = 256;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute((const void*)ffn_kernel<F32E, F32X>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_LAUNCH) == hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
    });
    if (!attr_ok) {
        wiw_set_error("hipFuncSetAttribute(ffn) failed");
        return WIW_ELAUNCH;
    }
    FfnArgs a;
    a.X = (const uint16_t*)X; a.W1 = (const uint16_t*)W1; a.b1 = b1; a.W2 = (const uint16_t*)W2; a.b2 = b2;
    a.rowvec = rowvec; a.res1 = (const uint16_t*)res1; a.res2 = (const uint16_t*)res2; a.out = (uint16_t*)out;
    a.M = (int)M; a.ldx = ldx; a.ldo = ldo; a.ldr1 = ldr1; a.ldr2 = ldr2; a.rowvec_ld = rowvec_ld;
    a.rows_per_vec = rows_per_vec > 0 ? rows_per_vec : 1; a.ln = ln;
    a.alpha = alpha; a.beta1 = beta1; a.beta2 = beta2; a.ln_eps = ln_eps; a.f32 = f32;
    a.out16 = (uint16_t*)out16; a.ldo16 = ldo16;
    const int tiles = (int)((M + BM - 1) / BM);
    const int grid = tiles < num_cu ? tiles : num_cu;
    hipLaunchKernelGGL((ffn_kernel<F32E, F32X>), dim3((unsigned)grid), dim3(512), SMEM_LAUNCH, (hipStream_t)stream, a);
    return wiw_check_launch("wiw_ffn_geglu_bf16");
}
}  // namespace

extern "C" int wiw_ffn_geglu_bf16(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                                  const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1,
                                  int ldr1, float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out,
                                  int ldo, int64_t M, int C_in, int hidden, int ln, float ln_eps) {
    return ffn_launch<false>(stream, X, ldx, W1, b1, W2, b2, rowvec, rowvec_ld, rows_per_vec, res1, ldr1, beta1, res2, ldr2, beta2,
                             alpha, out, ldo, M, C_in, hidden, ln, ln_eps, 0);
}

// ABI 12: the same operator on the fp32 residual stream — f32 bit 0: `out` is fp32, bit 1: res1, bit 2: res2 (include/wiw_svd.h)
extern "C" int wiw_ffn_geglu_f32stream(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                                       const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1,
                                       int ldr1, float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out,
                                       int ldo, int64_t M, int C_in, int hidden, int ln, float ln_eps, int f32) {
    WIW_REQUIRE(f32 >= 0 && f32 < 8, "ffn_geglu_f32stream: f32 is a 3-bit mask (out, res1, res2)");
    return ffn_launch<true>(stream, X, ldx, W1, b1, W2, b2, rowvec, rowvec_ld, rows_per_vec, res1, ldr1, beta1, res2, ldr2, beta2,
                            alpha, out, ldo, M, C_in, hidden, ln, ln_eps, f32);
}

// ABI 16: ... and with the fp32 stream on the INPUT side too — f32 bit 3: X is fp32 [M][ldx] (then ln != 0: the fused LayerNorm
// reads it; ldx in elements, a multiple of 4) — and an optional second output `out16` [M][ldo16], the 16-bit rounding of `out`
extern "C" int wiw_ffn_geglu_f32stream2(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                                        const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1,
                                        int ldr1, float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out,
                                        int ldo, int64_t M, int C_in, int hidden, int ln, float ln_eps, int f32, void* out16,
                                        int ldo16) {
    WIW_REQUIRE(f32 >= 0 && f32 < 16, "ffn_geglu_f32stream2: f32 is a 4-bit mask (out, res1, res2, X)");
    if (f32 & 8)
        return ffn_launch<true, true>(stream, X, ldx, W1, b1, W2, b2, rowvec, rowvec_ld, rows_per_vec, res1, ldr1, beta1, res2, ldr2,
                                      beta2, alpha, out, ldo, M, C_in, hidden, ln, ln_eps, f32 & 7, out16, ldo16);
    return ffn_launch<true>(stream, X, ldx, W1, b1, W2, b2, rowvec, rowvec_ld, rows_per_vec, res1, ldr1, beta1, res2, ldr2, beta2,
                            alpha, out, ldo, M, C_in, hidden, ln, ln_eps, f32, out16, ldo16);
}
