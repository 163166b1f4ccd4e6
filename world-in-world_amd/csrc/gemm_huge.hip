// 256 x 320 x 64 output tile ("HUGE") of the bf16 MFMA GEMM / implicit-GEMM convolution (gfx950).
//
// Why a second tile shape: in gemm.hip's 256x160 tile every wave owns 32 rows x 160 columns, i.e. per 32-deep k-step
// it reads 2 + 10 fragments from LDS for 20 MFMAs and the block moves 52 KiB of operands per 5.2 MFLOP through the
// LDS-DMA; ablations there show the operand path (DMA issue + ds_read + barriers) alone takes 1.45x the MFMA time.
// Here 8 waves form a 4 (M) x 2 (N) grid of 64 x 160 wave tiles on a 256 x 320 block tile:
//     per k-step and wave: 4 + 10 fragment reads for 40 MFMAs          (0.35 vs 0.60 ds_read_b128 per MFMA)
//     per K tile and block: 72 KiB of LDS-DMA for 10.5 MFLOP           (9 vs 7 DMA instructions per wave for 2x the MFMAs)
// The wave still owns 160 consecutive columns, so the GEGLU value | gate pairing (tiles of 160 = [80 | 80]) and the
// per-wave epilogue of gemm.hip carry over unchanged (4 passes of 16 rows instead of 2).
//
// Budget: 160 accumulator registers + 4 A and 5 B fragments (the 10 W fragments of a k-step are read in two halves)
// inside the 256-VGPR budget of 2 waves / SIMD; LDS ring = 2 stages x 72 KiB (144 of 160 KiB), one K tile of DMA in
// flight behind the one being consumed.
//
// Schedule per K tile (8 slots, a raw s_barrier between slots; waves 4..7 run one slot behind waves 0..3 so that on
// every SIMD one wave reads LDS / issues DMA while its partner issues 20 MFMAs back to back):
//     R(k0: A, W[0:5]) | M | R(k0: W[5:10]) | M | R(k1: A, W[0:5]) | M | R(k1: W[5:10]), confirm next tile | M
// DMA of K tile kt+1 (9 instructions per wave) is issued 2|2|2|2|1 in slots 0..4 of tile kt.
//
// Instantiations (template <MODE, GE, SK, HALO_, F32E>): GE = GEGLU epilogue (dense), SK = split-K pass 1 (raw fp32 slabs),
// HALO_ = 1 / 2: the 3x3 convolutions (stride 1, and behind a nearest x2 upsample) with a halo-staged A operand, 2 = plus the
// fused shortcut segment (round 4: 6 instead of 9 DMA instructions per wave and K tile, see the constants below), F32E = the
// fragment-layout epilogue of the fp32 residual stream.  Each is its own instantiation because this kernel lives at the
// 256-VGPR limit: a feature compiled into the common one costs every launch registers or scratch.
#include <stdlib.h>
#ifndef WIW_DMA_BURST
#define WIW_DMA_BURST 0   // 1: all DMA instructions of a K tile in ONE slot; 0: spread over five slots (measured: burst -2...-5 % on this tile, +2...+4 % on the 256x160 tile and the temporal block)
#endif
#ifndef WIW_ABLATE
#define WIW_ABLATE 0
#endif
#ifndef WIW_HUGE_PIPE
#define WIW_HUGE_PIPE 2   // 1: the free-running K loop of round 5 for every instantiation without a halo-staged A operand (below: PIPE); 2: also the halo-staged 3x3 convolutions
#endif
#if WIW_F16
#define WIW_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define WIW_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

#include <mutex>
#include <type_traits>

#include "common.h"
#ifndef WIW_GE_PK
#define WIW_GE_PK 1   // 1: GEGLU epilogue on packed pairs, breadth first (gelu_erf_pk).  Round 3 measured it 3 % SLOWER here with the
                      // A&S 7.1.26 erf (14 VALU: K = 640 595 vs 577 us, profiles/r06a_geglu_epilogue_ab.txt); with round 5's fitted
                      // sigmoid form (10 VALU, 13 issues per PAIR packed) it is FASTER: M = 64 512, K = 640: 470-476 -> 455-462 us,
                      // M = 16 128, K = 1280: 348-351 -> 344 us (profiles/r20l_geglu_epilogue_pk_ab.txt), bit-identical results; 0 = A/B
#endif

namespace {

constexpr int HM = 256, HN = 320, HK = 64;
constexpr int HA_BYTES = HM * HK * 2;            // 32768
constexpr int HB_BYTES = HN * HK * 2;            // 40960
constexpr int HSTAGE = HA_BYTES + HB_BYTES;      // 73728
constexpr int H_SMEM = 2 * HSTAGE;               // 147456
constexpr int HSTG_ROWB = 336;                   // bytes per staged bf16 row (160 cols + 16 B skew)
constexpr int HSTG_ROWB_G = 176;                 // GEGLU: 80 output cols + 16 B skew, 32 rows staged at once
constexpr int HSTG_WAVE = 32 * HSTG_ROWB_G;      // 5632 B per wave (plain: 16 rows x 336 B = 5376)
static_assert(16 * HSTG_ROWB <= HSTG_WAVE, "");
static_assert(8 * HSTG_WAVE <= HSTAGE, "epilogue staging must fit in one ring stage");
// HALO instantiation (3x3 convolution, round 4): the A operand is NOT fetched per tap.  A 256-row tile is R = 256 / Wd whole
// image rows; per 32-channel block the (R + 2) x (Wd + 2) pixel neighbourhood is staged ONCE ([halo pixel][32 ch = 64 B],
// pitch = Wd + 2 rounded up to 16 pixels, 16-byte chunks XOR-swizzled by (pixel >> 1) & 3: conflict-free for the 16-lane
// groups of ds_read_b128 at any tap offset) and the nine taps read it at pixel offsets dy * pitch + dx.  1/9 of the A
// bytes through the LDS-DMA (+ the halo): the ablation that fetched the A tile for the centre tap only ran the long-K
// convolutions 19-29 % faster (profiles/r10c_conv_a_dma_ablation.txt) — the fill rate of 128-byte row segments
// (25 B/clk/CU) is a first-order cost of this loop.  K walk: k = ((c / 32) * 9 + tap) * 32 + c % 32 (WIW_K_HALO32; a K tile
// is two taps of one 32-channel block).  LDS: 2 W stages of 40 KiB + 2 halo buffers of 36 KiB = 152 KiB.
constexpr int HALO_BUF = 36864;                  // 4 x 144 pixels (Wd = 128) / 6 x 80 pixels (Wd = 64) x 64 B
constexpr int HALO_SMEM = 2 * HB_BYTES + 2 * HALO_BUF;
static_assert(4 * HSTG_WAVE <= HB_BYTES && 4 * HSTG_WAVE <= HALO_BUF, "HALO epilogue staging: 4 waves per free region");

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

WIW_DEV void glds16(const char* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

__device__ uint4 g_dump_h[512 * 64];   // sink of the unconditional epilogue stores (see gemm.hip)

template <int N>
WIW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

WIW_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int V> using IC = std::integral_constant<int, V>;

// SK: split-K instantiation (raw fp32 slabs, no staged epilogue).  A separate template parameter because this kernel
// lives at the 256-VGPR limit: with the split path compiled into the common instantiation its spills doubled
// (80 -> 163 VGPRs) and every K >= 640 launch lost ~15 %.
// One K tile forward in the implicit-GEMM K walk (see the cursor declaration in the kernel): tap-major, or channel-block
// major under WIW_K_CMAJOR; the fused shortcut segment (tap 9, conv3x3 only) is walked last in both.
struct WiwKCur { int tap, cc; };
template <int MODE, int KT>
WIW_DEV WiwKCur wiw_advance_k(WiwKCur c, const int Ctot, const bool cmajor) {   // by value: by-reference cursors end up in scratch
    int ld_tap = c.tap, ld_cc = c.cc;
    constexpr int NT = MODE == WIW_A_CONV_T3 ? 3 : 9;
    if (MODE != WIW_A_DENSE && cmajor && !(MODE == WIW_A_CONV3X3 && ld_tap == 9)) {
        if (++ld_tap == NT) {
            ld_tap = 0;
            ld_cc += KT;
            if (MODE == WIW_A_CONV3X3 && ld_cc == Ctot) { ld_tap = 9; ld_cc = 0; }   // (only reached with a shortcut segment)
        }
    } else {
        ld_cc += KT;
        if (ld_cc == Ctot && !(MODE == WIW_A_CONV3X3 && ld_tap == 9)) { ld_cc = 0; ++ld_tap; }
    }
    return WiwKCur{ld_tap, ld_cc};
}

// HALO_: 0 = per-tap A tiles; 1 = halo-staged A operand; 2 = halo-staged + the fused shortcut segment behind the taps (a
// separate instantiation: the segment's per-row loader state costs the plain one 8 VGPRs and 36 bytes of scratch)
// F32E: the fp32 residual stream (WIW_EPI_OUT_F32 / _RES1_F32 / _RES2_F32, include/wiw_svd.h): no 16-bit staging — bias,
// per-frame vector and both residuals are added to the fp32 accumulators in the fragment layout (a lane holds 4 consecutive
// columns of a row: 16-byte fp32 / 8-byte 16-bit accesses in 64-byte row runs) and the sum is rounded once, or not at all.
// gemm.hip's 256x160 / 128x160 kernels have carried this epilogue since ABI 11; here it is its own instantiation for the
// same reason as there (compiled into the common one its residual registers cost every launch scratch).
// A1 (dense mode): ONE source (C2 = 0) and every row present (M % 256 = 0) — the A fetch then has no exec-masked row test
// and no concat-source test in front of it.  As a template argument: compiled out, the two tests cost the long-K dense
// launches 3-4 % (profiles/r10v_dense_a_fetch_tests_ablation.txt); behind a uniform run-time flag the gain was < 1 %.
template <int MODE, bool GE, bool SK, int HALO_ = 0, bool F32E = false, bool A1 = false>
__global__ __launch_bounds__(512, 2) void gemm_huge_kernel(const WiwGemmArgs p, const int stagger) {
    static_assert(!F32E || (!GE && !SK), "the fp32-stream epilogue: plain launches only");
    constexpr bool HALO = HALO_ != 0, HSEG = HALO_ == 2;
    // PIPE (round 5): the K loop without slots.  tools/ubench/mfma_dma.hip: a wave that interleaves its own ds_read_b128 with
    // its own MFMAs loses nothing (8 waves per CU, 14 fragment reads per 40 MFMAs: matrix pipe 0.94-1.00 busy), one s_barrier
    // per K tile costs 2-3 %, and the 9 LDS-DMA pieces per wave and K tile cost their issue time (64 cycles each) wherever
    // they are put — 0.80 busy for the whole loop against the 0.58-0.60 of the eight-slot schedule above, whose read slots
    // (9 / 5 KiB per wave) and seven pacing barriers per K tile are what the slots cost.  Every wave runs the same stream:
    // per 32-deep k-step  for j in 0..9 { wait W[j]; 4 MFMA (A[0..3] x W[j]); read W[j + 3] }  with the W fragments in a ring
    // of four and each A fragment of the next k-step read right behind its last MFMA; ONE barrier per K tile (iteration 7 of its
    // second k-step: every wave has issued its last read of the tile, every wave's share of the next tile has landed), the
    // DMA of tile kt + 2 may start right behind it.
    // HALO (round 5, second step): the same loop with the A fragments read from the halo buffers at the tap's offset; the
    // instantiation with the fused shortcut segment (HALO_ == 2: three kinds of K tile) keeps the slots.
    constexpr bool PIPE = WIW_HUGE_PIPE != 0 && HALO_ != 2 && (!HALO || WIW_HUGE_PIPE >= 2);
    static_assert(!HALO || ((MODE == WIW_A_CONV3X3 || MODE == WIW_A_CONV3X3_UP) && !GE && !SK) , "the halo-staged A operand is a plain 3x3 convolution");
    static_assert(!HSEG || MODE == WIW_A_CONV3X3, "the shortcut segment belongs to the stride-1 convolution");
    // UPH: nearest x2 upsample + 3x3 (WIW_A_CONV3X3_UP).  The staged image is the LOW-resolution input: tap (dy, dx) of output
    // pixel (y, x) reads input pixel ((y + dy) >> 1, (x + dx) >> 1), and the zero padding of the upsampled image is exactly
    // the out-of-range input coordinates — R / 2 + 2 input rows of Wd / 2 + 2 pixels per tile (a quarter of the pixels)
    constexpr bool UPH = HALO && MODE == WIW_A_CONV3X3_UP;
    constexpr int WSTAGE = HALO ? HB_BYTES : HSTAGE;          // bytes per ring stage
    constexpr int WOFF = HALO ? 0 : HA_BYTES;                 // offset of the W tile inside a stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;       // wave (wm, wn) owns rows wm*64.., columns wn*160.. of the block tile
    // Everything per-lane is re-derived from an OPAQUE copy of the lane id at the top of every tile and of every epilogue.
    // Without this hipcc hoists ~50 loop-invariant per-lane address registers out of the tile loop and, at the 256-VGPR
    // limit (160 accumulators), spills them around the K loop: 208-284 bytes of scratch per lane in round 3, whose
    // write-back was the 1.6-1.7x WRITE_SIZE excess of every instantiation of this kernel.
    int frow, fq, rsub, chunk;
    auto rederive = [&]() {
        asm volatile("" : "+v"(lane));
        frow = lane & 15; fq = lane >> 4; rsub = lane >> 3; chunk = (lane & 7) ^ (rsub & 7);
    };
    rederive();

    // ---- tile schedule: per-XCD sm x sn super-tiles (identical to gemm.hip, with this tile shape)
    const int Nt = (p.N + HN - 1) / HN;
    // split-K (p.splitk = S > 1), as in gemm.hip: K range ks of tile row tr is schedule row ks * Mt1 + tr and writes raw
    // fp32 sums to slab ks of the workspace (p.out here)
    const int S = SK ? p.splitk : 1;
    const int Mt1 = (p.M + HM - 1) / HM;
    const int Mt = Mt1 * S;
    const int total = Mt * Nt;
    const int nb = gridDim.x;
    const bool super = (nb & 7) == 0 && nb >= 64;
    const int bpx = nb >> 3;
    int sn = 1;
    {
        int64_t best = -1;
        for (int c = 1; c <= 8; c *= 2) {
            if (c > bpx || (c > 1 && c > Nt)) break;
            const int cm = bpx / c;
            const int64_t slots = (int64_t)((Nt + c - 1) / c) * c * ((Mt + cm - 1) / cm) * cm;
            if (best < 0 || slots <= best) { best = slots; sn = c; }
        }
    }
    {   // tuning knob WIW_GEMM_SN=1|2|4|8 (bits 8..11 of `stagger`): force the width of the per-XCD super-tile
        const int sn_force = (stagger >> 8) & 15;
        if (sn_force > 0 && sn_force <= bpx && sn_force <= Nt) sn = sn_force;
    }
    const int sm = super ? bpx / sn : 1;
    const int SNt = (Nt + sn - 1) / sn, SMt = (Mt + sm - 1) / sm;
    const int n_super = SNt * SMt;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    int q = xcd, t_end = 0;
    auto next_tile = [&](int& pos) -> int {
        if (!super) return pos < t_end ? pos : -1;
        for (; pos < n_super; pos += 8) {
            const int tm = (pos / SNt) * sm + jx / sn, tn = (pos % SNt) * sn + jx % sn;
            if (tm < Mt && tn < Nt) return tm * Nt + tn;
        }
        return -1;
    };
    if (!super) {
        const int qq = nb >> 3, r = nb & 7, idx = blockIdx.x >> 3;
        const int lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
        q = (int)(((int64_t)lb * total) / nb);
        t_end = (int)(((int64_t)(lb + 1) * total) / nb);
    }
    int t = next_tile(q);
    if (t < 0) return;
    if ((stagger & 255) > 0) {   // de-phase the blocks (tuning knob WIW_GEMM_STAGGER): see gemm.hip
        const int phase = (blockIdx.x >> 3) & 7;
        for (int i = 0; i < phase * (stagger & 255); ++i) __builtin_amdgcn_s_sleep(16);   // 1024 cycles per iteration
    }

    // ---- loader state
    const char* const Ab = (const char*)p.A;
    const char* const A2b = (const char*)p.A2;
    const char* const A3b = (const char*)p.A3;
    const char* const zeros = (const char*)p.zeros;
    const int HW = p.H * p.Wd;
    const int Ctot = MODE == WIW_A_DENSE ? p.C1 + p.C2 : p.C1;   // channels per tap (the shortcut segment is tap 9)
    const int nk = p.K / HK / S;       // K tiles per output tile (of one split range)
    const int nk_w = p.K / HK;         // K tiles per 8-row block of a tiled W

    int a_m[4];
    bool a_ok[4];
    int a_fb[4], a_y[4], a_x[4];
    const char* w_row[5];
    // WIW_W_TILED: W pre-tiled by the host into 1-KiB blocks [n / 8][k tile][8 rows x 128 B, chunks pre-swizzled]: one
    // DMA instruction then reads ONE contiguous KiB (63 B/clk/CU) instead of 8 row segments K*2 bytes apart (25 B/clk/CU,
    // tools/ubench/lds_fill.hip) — the LDS image is byte-identical
    const bool w_tiled = (p.epilogue & WIW_W_TILED) != 0;
    const int64_t w_kstep = w_tiled ? 1024 : HK * 2;
    int ld_tap = 0, ld_cc = 0, ld_kt = 0;
    // K walk of the implicit GEMM.  Tap-major (k = tap * Ctot + c) or, with WIW_K_CMAJOR, channel-block-major
    // (k = ((c / 64) * taps + tap) * 64 + c % 64): the taps of one 64-channel block are consecutive K tiles and re-read the
    // same activation window while it is still in the XCD's L2 (include/wiw_svd.h).  The shortcut segment (tap 9) is last.
    constexpr int NTAPS = MODE == WIW_A_CONV_T3 ? 3 : 9;
    const bool cmajor = MODE != WIW_A_DENSE && (p.epilogue & WIW_K_CMAJOR) != 0;

    auto setup_loader = [&](int tile) {
        const int m0 = (SK ? (tile / Nt) % Mt1 : tile / Nt) * HM, n0 = (tile % Nt) * HN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + (wave * 4 + i) * 8 + rsub;
            a_m[i] = m;
            a_ok[i] = m < p.M;
            a_fb[i] = 0; a_y[i] = 0; a_x[i] = 0;
            if (MODE == WIW_A_CONV3X3 || MODE == WIW_A_CONV3X3_S2 || MODE == WIW_A_CONV3X3_S2P || MODE == WIW_A_CONV3X3_UP) {
                const int n = m / HW, rem = m - n * HW;
                a_y[i] = rem / p.Wd;
                a_x[i] = rem - a_y[i] * p.Wd;
                a_fb[i] = (MODE == WIW_A_CONV3X3) ? 0      // unused: the tap row is a_m + dy * Wd + dx
                          : ((MODE == WIW_A_CONV3X3_S2 || MODE == WIW_A_CONV3X3_S2P) ? n * HW * 4 : n * (HW >> 2));
            } else if (MODE == WIW_A_CONV_T3) {
                a_y[i] = (m / HW) % p.T;
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (w_tiled) {
                int blk = (n0 >> 3) + wave * 5 + i;
                const int last = ((p.N + 7) >> 3) - 1;
                blk = blk < last ? blk : last;       // blocks past N re-read the last one (computed, never stored)
                w_row[i] = (const char*)p.W + (int64_t)blk * nk_w * 1024 + lane * 16;
            } else {
                int n = n0 + (wave * 5 + i) * 8 + rsub;
                n = n < p.N ? n : p.N - 1;   // clamped rows are computed but never stored
                w_row[i] = (const char*)p.W + ((int64_t)n * p.K + chunk * 8) * 2;
            }
        }
    };

    static_assert(!A1 || (MODE == WIW_A_DENSE && !SK), "A1 is a dense-mode loader variant");
    auto a_src = [&](int i, int tap, int cc) -> const char* {
        if (MODE == WIW_A_DENSE) {
            if constexpr (A1) return Ab + ((int64_t)a_m[i] * p.C1 + cc + chunk * 8) * 2;
            if (!a_ok[i]) return zeros;
            if (cc < p.C1) return Ab + ((int64_t)a_m[i] * p.C1 + cc + chunk * 8) * 2;
            return A2b + ((int64_t)a_m[i] * p.C2 + (cc - p.C1) + chunk * 8) * 2;
        } else if (MODE == WIW_A_CONV_T3) {
            const int tt = a_y[i] + tap - 1;
            if (!a_ok[i] || (unsigned)tt >= (unsigned)p.T) return zeros;
            return Ab + (((int64_t)a_m[i] + (int64_t)(tap - 1) * HW) * p.C1 + cc + chunk * 8) * 2;
        } else {
            if (MODE == WIW_A_CONV3X3 && tap == 9) {   // fused 1x1 shortcut: dense segment concat(A2, A3) after the taps
                if (!a_ok[i]) return zeros;
                if (cc < p.C2) return A2b + ((int64_t)a_m[i] * p.C2 + cc + chunk * 8) * 2;
                return A3b + ((int64_t)a_m[i] * p.C3 + (cc - p.C2) + chunk * 8) * 2;
            }
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            int row;
            bool ok;
            if (MODE == WIW_A_CONV3X3) {
                const int iy = a_y[i] + dy, ix = a_x[i] + dx;
                ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                row = a_m[i] + dy * p.Wd + dx;     // = frame base + iy * Wd + ix: no per-row frame base to keep live
            } else if (MODE == WIW_A_CONV3X3_S2 || MODE == WIW_A_CONV3X3_S2P) {
                constexpr int o = (MODE == WIW_A_CONV3X3_S2P) ? 1 : 0;
                const int iy = 2 * a_y[i] + dy + o, ix = 2 * a_x[i] + dx + o;
                ok = (unsigned)iy < (unsigned)(2 * p.H) && (unsigned)ix < (unsigned)(2 * p.Wd);
                row = a_fb[i] + iy * (2 * p.Wd) + ix;
            } else {
                const int uy = a_y[i] + dy, ux = a_x[i] + dx;
                ok = (unsigned)uy < (unsigned)p.H && (unsigned)ux < (unsigned)p.Wd;
                row = a_fb[i] + (uy >> 1) * (p.Wd >> 1) + (ux >> 1);
            }
            if (!a_ok[i] || !ok) return zeros;
            return Ab + ((int64_t)row * p.C1 + cc + chunk * 8) * 2;
        }
    };

    char* const halo0 = smem + 2 * HB_BYTES;                     // HALO: the two halo buffers behind the two W stages
    int h_m0 = 0;                                                // HALO: first row of the tile in the K loop
    const int h_kc = HSEG ? (9 * p.C1) / HK : 0x7fffffff;       // HALO: K tiles of the taps; the shortcut segment (C2 + C3) follows
    // the 9 DMA instructions of a K tile in five parts: A0 A1 | A2 A3 | W0 W1 | W2 W3 | W4 (+ advance the K cursor)
    auto issue_part = [&](int stage, auto part_tag, auto fseg_tag) {
        constexpr int part = decltype(part_tag)::value;
        constexpr bool fseg = decltype(fseg_tag)::value != 0;     // HALO_ == 2: the tile being fetched is a shortcut-segment tile
        // HALO: only the K tiles of the shortcut segment have an A tile of their own; it goes to halo buffer
        // (ld_kt - h_kc) & 1 (ld_kt = the K tile being fetched), free since the taps' block nb - 2 + that parity was last read
        char* sA = HALO ? halo0 + ((ld_kt - h_kc) & 1) * HALO_BUF + wave * 4 * 1024 : smem + stage * HSTAGE + wave * 4 * 1024;
        char* sB = smem + stage * WSTAGE + WOFF + wave * 5 * 1024;
        if constexpr (HALO && part < 2) {
            if constexpr (fseg) {
                // the row / chunk of this lane from the lane id (the per-row loader state of the other instantiations would
                // stay live across the K loop: 8 VGPRs this kernel does not have); M % 256 == 0: every row exists
                const int cc = (ld_kt - h_kc) * HK, ck = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int64_t m = h_m0 + (wave * 4 + 2 * part + u) * 8 + (lane >> 3);
                    const char* src = cc < p.C2 ? A2b + (m * p.C2 + cc + ck) * 2 : A3b + (m * p.C3 + (cc - p.C2) + ck) * 2;
                    glds16(src, sA + (2 * part + u) * 1024);
                }
            }
        } else if constexpr (part < 2) {
#if WIW_ABLATE == 30   // timing experiment only (wrong results): the A tile of a 3x3 convolution is fetched for the centre tap
                       // alone = the LDS-DMA instruction count of a halo-staged A operand (1 of 9 A tiles per channel block)
            if (MODE != WIW_A_CONV3X3 || ld_tap == 4 || ld_tap == 9)
#endif
            {
            glds16(a_src(2 * part, ld_tap, ld_cc), sA + (2 * part) * 1024);
            glds16(a_src(2 * part + 1, ld_tap, ld_cc), sA + (2 * part + 1) * 1024);
            }
        } else if constexpr (part < 4) {
            constexpr int i = 2 * (part - 2);
            glds16(w_row[i] + (int64_t)ld_kt * w_kstep, sB + i * 1024);
            glds16(w_row[i + 1] + (int64_t)ld_kt * w_kstep, sB + (i + 1) * 1024);
        } else {
            glds16(w_row[4] + (int64_t)ld_kt * w_kstep, sB + 4 * 1024);
            ++ld_kt;
            { const WiwKCur kc = wiw_advance_k<MODE, HK>(WiwKCur{ld_tap, ld_cc}, Ctot, cmajor); ld_tap = kc.tap; ld_cc = kc.cc; }
        }
    };
    auto issue_all = [&](int stage) {
        issue_part(stage, IC<0>{}, IC<0>{}); issue_part(stage, IC<1>{}, IC<0>{}); issue_part(stage, IC<2>{}, IC<0>{});
        issue_part(stage, IC<3>{}, IC<0>{}); issue_part(stage, IC<4>{}, IC<0>{});
    };

    // ---- HALO: geometry (uniform), the DMA instruction -> halo position map of this wave, the tap cursor
    const int h_R = HALO ? 256 / p.Wd : 1;                       // image rows per tile
    const int h_ipr = HALO ? ((UPH ? p.Wd >> 1 : p.Wd) + 2 + 15) >> 4 : 1;   // DMA instructions (16 pixels x 64 B) per halo row
    const int h_P = h_ipr * 16;                                  // halo pitch in pixels
    // halo rows: one above, the R rows, one below — and, when tiles straddle frames (H * Wd % 256 != 0), a ZERO row between
    // the last row of a frame and the first row of the next (what both frames' taps across that edge must read)
    const int h_ninstr = HALO ? (UPH ? (h_R >> 1) + 2 : h_R + 2 + (HW % HM != 0 ? 1 : 0)) * h_ipr : 0;    // <= 36
    const int h_nblk = p.C1 >> 5;                                // 32-channel blocks
    int h_hy[5], h_jx[5];                                        // instruction j = wave + 8 i of a block: halo row, 16-pixel column
#pragma unroll
    for (int i = 0; i < 5; ++i) { const int j = wave + 8 * i; h_hy[i] = j / h_ipr; h_jx[i] = j - h_hy[i] * h_ipr; }
    int h_tap = 0, h_blk = 0, h_next = 1, h_tile = 0;
    int h_S[4] = {0, 0, 0, 0};                                   // halo pixel of row 0 of this wave's 16-row block mi at tap (0, 0)
    // tile geometry: y0 = image row of the tile's first row, bpos = tile rows that belong to its frame (>= R: all of them)
    auto tile_y0 = [&](int m0t) { return (m0t % HW) / p.Wd; };
    // instruction i of this wave for 32-channel block b of the tile whose first row is m0t (first image row y0t)
    auto issue_halo = [&](int m0t, int y0t, int b, auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const int j = wave + 8 * i;
        if (j >= h_ninstr) return;
        if constexpr (UPH) {     // input rows y0 / 2 - 1 ... of the tile's frame (tiles are whole rows of one frame, y0 even)
            const int iy = (y0t >> 1) - 1 + h_hy[i];
            const int ix = h_jx[i] * 16 + (lane >> 2) - 1;
            const bool ok = (unsigned)iy < (unsigned)(p.H >> 1) && (unsigned)ix < (unsigned)(p.Wd >> 1);
            const int csrc = (lane & 3) ^ ((lane >> 3) & 3);
            const int64_t fb = (int64_t)(m0t / HW) * (HW >> 2);
            const char* src = ok ? Ab + ((fb + iy * (p.Wd >> 1) + ix) * p.C1 + b * 32 + csrc * 8) * 2 : zeros;
            glds16(src, halo0 + (b & 1) * HALO_BUF + j * 1024);
            return;
        }
        const int bpos = p.H - y0t;                              // first tile row of the NEXT frame (if < R)
        const int hy = h_hy[i];
        const bool split = bpos < h_R;
        const bool gap = split && hy == bpos + 1;
        const int r = hy - 1 - ((split && hy > bpos + 1) ? 1 : 0);           // tile row this halo row shows (-1 / R: the outer halo)
        const int y = (split && r >= bpos) ? r - bpos : y0t + r;              // its image row in its frame
        const int ix = h_jx[i] * 16 + (lane >> 2) - 1;
        const bool ok = !gap && (unsigned)y < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
        const int csrc = (lane & 3) ^ ((lane >> 3) & 3);        // LDS chunk (lane & 3) of pixel hp holds source chunk ^ ((hp >> 1) & 3)
        const char* src = ok ? Ab + (((int64_t)m0t + r * p.Wd + ix) * p.C1 + b * 32 + csrc * 8) * 2 : zeros;
        glds16(src, halo0 + (b & 1) * HALO_BUF + j * 1024);
    };
    auto issue_halo_all = [&](int m0t, int y0t, int b) {
        issue_halo(m0t, y0t, b, IC<0>{}); issue_halo(m0t, y0t, b, IC<1>{}); issue_halo(m0t, y0t, b, IC<2>{});
        issue_halo(m0t, y0t, b, IC<3>{}); issue_halo(m0t, y0t, b, IC<4>{});
    };
    // DMA of slot `sl` (0..4) of a K tile: the five parts of the next K tile.  HALO, taps: its 5 W instructions in slots 0..2
    // and, in the tile where block h_next may be fetched (hdo), that block's <= 5 halo instructions in slots 2..4 (measured
    // against W in the wave's read slots only and against one burst in slot 0: 1-3 % slower, profiles/r10f_halo_sched_probe.txt);
    // HALO, shortcut segment (the tile being fetched is ld_kt >= h_kc): A | A | W | W | W like the other instantiations
    auto issue_slot = [&](int si, auto slot_tag, auto fseg_tag, bool more, bool hdo, int m0t, int y0t) {
        constexpr int sl = decltype(slot_tag)::value;
        constexpr bool fseg = decltype(fseg_tag)::value != 0;
        if constexpr (!HALO) {
            if (more) issue_part(si, IC<sl>{}, IC<0>{});
        } else {
            if constexpr (fseg) {
                if (more) issue_part(si, IC<sl>{}, IC<1>{});
            } else if (more) {
                if constexpr (sl == 0) issue_part(si, IC<2>{}, IC<0>{});
                else if constexpr (sl == 1) issue_part(si, IC<3>{}, IC<0>{});
                else if constexpr (sl == 2) issue_part(si, IC<4>{}, IC<0>{});
            }
            if (hdo) {
                if constexpr (sl == 2) issue_halo(m0t, y0t, h_next, IC<0>{});
                else if constexpr (sl == 3) { issue_halo(m0t, y0t, h_next, IC<1>{}); issue_halo(m0t, y0t, h_next, IC<2>{}); }
                else if constexpr (sl == 4) { issue_halo(m0t, y0t, h_next, IC<3>{}); issue_halo(m0t, y0t, h_next, IC<4>{}); }
            }
        }
    };

    f32x4 acc[4][10];
    bf16x8 fa[4], fb[5];
    auto read_a_halo = [&](int kt, int kk, auto cseg_tag) {
        if constexpr (decltype(cseg_tag)::value != 0) {   // shortcut segment: a [256][64] tile in halo buffer (kt - h_kc) & 1, laid out like a ring stage's A part
            const int sw = ((kk * 4 + fq) ^ (lane & 7)) << 4;
            const char* sA = halo0 + ((kt - h_kc) & 1) * HALO_BUF + (wm * 64 + frow) * 128 + sw;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi] = *(const bf16x8*)(sA + mi * 2048);
            return;
        }
        // the A fragments of the next 32-deep k-step: tap h_tap of block h_blk.  h_S, the pitch and the 16-pixel columns are
        // multiples of 16 (+1), so the swizzle phase of pixel h_S + tap offset + frow depends on dx and the lane only
        const int dy = h_tap / 3 - 1, dx = h_tap - (h_tap / 3) * 3 - 1;
        if constexpr (UPH) {
            // Wd >= 64: the wave's 64 rows lie in ONE output row r = h_S[1]; block mi starts at input column h_S[0] + 8 mi.
            // Lanes 2k, 2k + 1 (shifted by dx) read the same input pixel: same address, a broadcast
            const int fx = (frow + dx) >> 1;                                            // -1 ... 8
            const int lp = fx * 64 + ((fq ^ (((1 + fx) >> 1) & 3)) << 4);
            const char* sA = halo0 + (h_blk & 1) * HALO_BUF + ((((h_S[1] + dy) >> 1) + 1) * h_P + h_S[0]) * 64 + lp;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[mi] = *(const bf16x8*)(sA + mi * 512);
            if (++h_tap == 9) { h_tap = 0; ++h_blk; }
            return;
        }
        const int lp = frow * 64 + ((fq ^ (((1 + dx + frow) >> 1) & 3)) << 4);
        const char* sA = halo0 + (h_blk & 1) * HALO_BUF + (dy * h_P + dx) * 64 + lp;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = *(const bf16x8*)(sA + h_S[mi] * 64);
        if (++h_tap == 9) { h_tap = 0; ++h_blk; }
    };
    auto read_a = [&](int stage, int kk, int kt, auto cseg_tag) {
        if constexpr (HALO) { read_a_halo(kt, kk, cseg_tag); return; }
        const int sw = ((kk * 4 + fq) ^ (lane & 7)) << 4;
        const char* sA = smem + stage * HSTAGE + (wm * 64 + frow) * 128 + sw;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = *(const bf16x8*)(sA + mi * 2048);
    };
    auto read_b = [&](int stage, int kk, auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        const int sw = ((kk * 4 + fq) ^ (lane & 7)) << 4;
        const char* sB = smem + stage * WSTAGE + WOFF + (wn * 160 + h * 80 + frow) * 128 + sw;
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = *(const bf16x8*)(sB + j * 2048);
    };
    auto mma = [&](auto h_tag) {
        constexpr int h = decltype(h_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)   // swapped operands: lane gets n = 16*(5h+j) + 4*fq + r, m = 16*mi + frow
                acc[mi][h * 5 + j] = WIW_MFMA(fb[j], fa[mi], acc[mi][h * 5 + j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool lag = wave >= 4;

    const bool scale_acc = p.alpha != 1.0f;
    const int n_valid = GE ? p.n_out : p.N;
    const uint16_t* r1 = (const uint16_t*)p.res1;
    const uint16_t* r2 = (const uint16_t*)p.res2;

    auto reset_loader = [&](int tile) {   // K position of the loader at the start of `tile` (its split range)
        ld_kt = SK ? ((tile / Nt) / Mt1) * nk : 0;
        const int k0 = ld_kt * HK;
        if (MODE == WIW_A_DENSE) { ld_tap = 0; ld_cc = k0; }
        else {
            if (cmajor) {
                const int kt0 = k0 / HK, nblk = Ctot / HK;
                if (kt0 < NTAPS * nblk) { ld_tap = kt0 % NTAPS; ld_cc = (kt0 / NTAPS) * HK; }
                else { ld_tap = 9; ld_cc = (kt0 - NTAPS * nblk) * HK; }
                return;
            }
            int tp = k0 / Ctot;
            if (MODE == WIW_A_CONV3X3 && tp > 9) tp = 9;
            ld_tap = tp; ld_cc = k0 - tp * Ctot;
        }
    };

    // ---- prologue: K tile 0 of the first output tile
    setup_loader(t);
    reset_loader(t);
    int st_c = 0;
    int pending_stores = 0;   // 0 / 10 (GEGLU) / 24
    issue_all(0);
    if constexpr (HALO) { const int m0f = (t / Nt) * HM; issue_halo_all(m0f, tile_y0(m0f), 0); }

    while (t >= 0) {
        const int tile_n = t % Nt;
        const int m0 = (SK ? (t / Nt) % Mt1 : t / Nt) * HM;
        const int64_t out_slab = SK ? (int64_t)((t / Nt) / Mt1) * p.M * p.ldo : 0;
        rederive();
        setup_loader(t);
        const int y0 = HALO ? tile_y0(m0) : 0;
        h_m0 = m0;
        h_tap = 0; h_blk = 0; h_next = 1; h_tile = 0;       // (HALO) block 0 was fetched with K tile 0; block 1 may go at once
        if constexpr (UPH) {
            const int rr = wm * 64, r = rr / p.Wd;
            h_S[0] = ((rr - r * p.Wd) >> 1) + 1;        // halo column of the wave's first pixel
            h_S[1] = r;                                  // its output row inside the tile
        } else if constexpr (HALO) {
            const int bpos = p.H - y0;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int rr = wm * 64 + mi * 16, r = rr / p.Wd, c = rr - r * p.Wd;
                h_S[mi] = (r + 1 + ((bpos < h_R && r >= bpos) ? 1 : 0)) * h_P + c + 1;
            }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        // K tile 0 was issued BEFORE the previous tile's epilogue stores: it has landed once at most those stores are
        // outstanding (in-order retirement; their number is a compile-time constant because they are unconditional)
        if (pending_stores == 24) wait_vmcnt<24>();
        else if (pending_stores == 10) wait_vmcnt<10>();
        else wait_vmcnt<0>();

        // Sync proof (same as gemm.hip with 8 slots): the lagging group's local barrier b is the leading group's b+1.
        //   * DMA into the other stage (held tile kt-1) starts after local barrier 8kt: every wave has passed its
        //     barrier 8kt-1, i.e. finished slot 8(kt-1)+6 — the last read of that stage;
        //   * tile kt+1 is first read in the leading group's slot 8(kt+1) = after the lagging group's barrier 8kt+7,
        //     so every wave confirms its DMA share before its local barrier 8kt+7 (end of slot 6).
        if constexpr (PIPE) {
            // ---- the free-running K loop (see PIPE above).  Fragment registers: A of the running / the next k-step (pa[kk & 1]),
            // W in a ring of four (fragment w of k-step kk lives in slot (2 kk + w) & 3: 20 fragments per K tile, so the phase
            // repeats every tile); reads run three fragments ahead of the MFMAs.  Every ds_read and every MFMA is a volatile asm
            // statement: the order below IS the instruction order, and the counted waits say which reads may still be in flight.
#define HP_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define HP_MMA(c, w_, a_) asm volatile(WIW_MFMA_ASM " %0, %1, %2, %0" : "+v"(c) : "v"(w_), "v"(a_))
#define HP_WAIT(n, r) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(n))
            // The MFMAs below are asm statements: hipcc does not pad hazards for them.  The only VALU writes they can meet are the
            // zeros of the accumulators, which the compiler materialises as late as it can — found as ONE stale register of ONE
            // accumulator (a v_mov_b64 of zeros directly in front of its first MFMA, profiles/r16_pipe_kloop.txt).  These two
            // statements take all 160 registers as operands: the zeros exist before them, the s_nop is the padding.
#define HP_A10(m) "+v"(acc[m][0]), "+v"(acc[m][1]), "+v"(acc[m][2]), "+v"(acc[m][3]), "+v"(acc[m][4]), "+v"(acc[m][5]), "+v"(acc[m][6]), "+v"(acc[m][7]), "+v"(acc[m][8]), "+v"(acc[m][9])
            asm volatile("s_nop 7" : HP_A10(0), HP_A10(1));
            asm volatile("s_nop 7" : HP_A10(2), HP_A10(3));
            slot_barrier();        // K tile 0 (confirmed above by every wave) is visible; every wave has left the previous epilogue's staging area
            bf16x8 pa[4], pb[4];
            unsigned ra[2], rb[2];     // this lane's fragment addresses in the stage of the running tile, k-step 0 / 1 (HALO: ra unused)
            int nd;                    // byte distance to the other stage
            {
                const unsigned ro0 = (unsigned)(frow * 128 + ((fq ^ (lane & 7)) << 4)), ro1 = ro0 ^ 64u;
                const unsigned sa = (unsigned)(uintptr_t)(smem + st_c * HSTAGE) + (unsigned)(wm * 8192);
                const unsigned sb = (unsigned)(uintptr_t)(smem + st_c * WSTAGE + WOFF) + (unsigned)(wn * 20480);
                ra[0] = sa + ro0; ra[1] = sa + ro1; rb[0] = sb + ro0; rb[1] = sb + ro1;
                nd = st_c ? -WSTAGE : WSTAGE;
            }
            // HALO: the A fragments of k-step s are tap s % 9 of 32-channel block s / 9, read from halo buffer (s / 9) & 1 at the
            // tap's pixel offset (read_a_halo above).  (p_tap, p_blk) = the k-step whose fragments are read NEXT.
            int p_tap = 0, p_blk = 0;
            unsigned hS64[4] = {0u, 0u, 0u, 0u};
            if constexpr (HALO && !UPH) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) hS64[mi] = (unsigned)(h_S[mi] * 64);
            }
            auto halo_a = [&]() -> unsigned {      // this lane's address of fragment mi = 0 of k-step (p_tap, p_blk); advances the cursor
                const int dy = p_tap / 3 - 1, dx = p_tap - (p_tap / 3) * 3 - 1;
                unsigned a;
                if constexpr (UPH) {
                    const int fx = (frow + dx) >> 1;
                    const int lp = fx * 64 + ((fq ^ (((1 + fx) >> 1) & 3)) << 4);
                    a = (unsigned)(uintptr_t)(halo0 + (p_blk & 1) * HALO_BUF) + (unsigned)(((((h_S[1] + dy) >> 1) + 1) * h_P + h_S[0]) * 64 + lp);
                } else {
                    const int lp = frow * 64 + ((fq ^ (((1 + dx + frow) >> 1) & 3)) << 4);
                    a = (unsigned)(uintptr_t)(halo0 + (p_blk & 1) * HALO_BUF) + (unsigned)((dy * h_P + dx) * 64 + lp);
                }
                if (++p_tap == 9) { p_tap = 0; ++p_blk; }
                return a;
            };
            // prologue, in the order the steady state leaves its reads: W[0] W[1] A[0..3] W[2]
            HP_DSR(pb[0], rb[0], 0); HP_DSR(pb[1], rb[0], 2048);
            if constexpr (!HALO) {
                HP_DSR(pa[0], ra[0], 0); HP_DSR(pa[1], ra[0], 2048); HP_DSR(pa[2], ra[0], 4096); HP_DSR(pa[3], ra[0], 6144);
            } else if constexpr (UPH) {
                const unsigned a0 = halo_a();
                HP_DSR(pa[0], a0, 0); HP_DSR(pa[1], a0, 512); HP_DSR(pa[2], a0, 1024); HP_DSR(pa[3], a0, 1536);
            } else {
                const unsigned a0 = halo_a();
                const unsigned a00 = a0 + hS64[0], a01 = a0 + hS64[1], a02 = a0 + hS64[2], a03 = a0 + hS64[3];
                HP_DSR(pa[0], a00, 0); HP_DSR(pa[1], a01, 0); HP_DSR(pa[2], a02, 0); HP_DSR(pa[3], a03, 0);
            }
            HP_DSR(pb[2], rb[0], 4096);
            // one k-step, KK its parity.  Program order of the reads:  iteration j < 7: W[j + 3];  iteration 7: W'[0] (behind the
            // tile's barrier when the next k-step is the next tile's);  iterations 8 and 9, merged (4 x { MFMA(8, mi); MFMA(9, mi);
            // A'[mi] }): W'[1] behind the first MFMA, W'[2] at the end — ONE set of A registers, each reloaded right behind its last
            // two MFMAs.  The last K tile of an output tile runs the same code (ONE copy of the loop body: with a second flavour for
            // it hipcc kept the accumulators of one flavour in scratch): its reads ahead fetch stale bytes of the other stage that
            // nothing uses, its barrier is one barrier more.
            // DMA: K tile kt + 1 goes into the other stage during the first k-step of tile kt (five parts, iterations 0 2 4 6 8/9).
            // Fetching K tile kt + 2 right behind tile kt's barrier instead (a whole K tile of latency cover) measured no faster
            // than the slot loop: a burst of DMA issue behind the barrier is the worst place for it (profiles/r16f_pipe_ab.txt).
            auto pipe_step = [&](auto kk_tag, const bool dma1, const int si, const bool hdo, const int m0t, const int y0t) {
                constexpr int KK = decltype(kk_tag)::value;
                const unsigned ra_n = HALO ? 0u : (KK == 0 ? ra[1] : ra[0] + (unsigned)nd);      // the next k-step's fragment addresses
                const unsigned rb_n = KK == 0 ? rb[1] : rb[0] + (unsigned)nd;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bf16x8& w_now = pb[(2 * KK + j) & 3];
                    // in flight behind what iteration j needs — j = 0: A[mi] is followed by A[mi + 1 ..], W[2] (and W[3], read
                    // behind the first MFMA); j >= 1: the two W fragments read since
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w_now), "+v"(pa[0]));
                    else HP_WAIT(2, w_now);
                    if (KK == 1 && j == 7) {
                        // the tile's barrier: my share of K tile kt + 1 has landed, my last read of this tile is done
                        // (lgkmcnt(0) inside slot_barrier) — behind it every wave may read the other stage and DMA into this one
                        wait_vmcnt<0>();
                        slot_barrier();
                    }
                    HP_MMA(acc[0][j], w_now, pa[0]);
                    {      // W[j + 3], or W'[0] of the next k-step
                        bf16x8& w_new = pb[(2 * KK + j + 3) & 3];
                        if (j < 7) HP_DSR(w_new, rb[KK], (j + 3) * 2048);
                        else HP_DSR(w_new, rb_n, 0);
                    }
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pa[1]));
                    HP_MMA(acc[1][j], w_now, pa[1]);
                    if (!HALO && KK == 0 && (j & 1) == 0 && dma1) {      // the 9 DMA instructions of K tile kt + 1, in five parts (the fifth below)
                        if (j == 0) issue_part(si, IC<0>{}, IC<0>{});
                        else if (j == 2) issue_part(si, IC<1>{}, IC<0>{});
                        else if (j == 4) issue_part(si, IC<2>{}, IC<0>{});
                        else issue_part(si, IC<3>{}, IC<0>{});
                    }
                    if constexpr (HALO) {      // 5 W instructions of K tile kt + 1 (2 | 2 | 1) and, when a halo block may be fetched, its <= 5
                        if (KK == 0 && dma1) {
                            if (j == 0) issue_part(si, IC<2>{}, IC<0>{});
                            else if (j == 2) issue_part(si, IC<3>{}, IC<0>{});
                            else if (j == 4) issue_part(si, IC<4>{}, IC<0>{});
                        }
                        if (KK == 0 && hdo) {
                            if (j == 5) issue_halo(m0t, y0t, h_next, IC<0>{});
                            else if (j == 6) issue_halo(m0t, y0t, h_next, IC<1>{});
                            else if (j == 7) issue_halo(m0t, y0t, h_next, IC<2>{});
                        }
                    }
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(pa[2]));
                    HP_MMA(acc[2][j], w_now, pa[2]);
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(pa[3]));
                    HP_MMA(acc[3][j], w_now, pa[3]);
                }
                {   // iterations 8 and 9
                    bf16x8& w8 = pb[(2 * KK + 8) & 3];
                    bf16x8& w9 = pb[(2 * KK + 9) & 3];
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w8), "+v"(w9));      // W'[0] may be in flight
                    unsigned an0 = ra_n, an1 = ra_n, an2 = ra_n, an3 = ra_n;      // HALO: the four fragment addresses of the next k-step
                    if constexpr (HALO) {
                        const unsigned a0 = halo_a();
                        if constexpr (UPH) { an0 = an1 = an2 = an3 = a0; }
                        else { an0 = a0 + hS64[0]; an1 = a0 + hS64[1]; an2 = a0 + hS64[2]; an3 = a0 + hS64[3]; }
                    }
                    constexpr int AO1 = HALO ? (UPH ? 512 : 0) : 2048, AO2 = 2 * AO1, AO3 = 3 * AO1;
                    HP_MMA(acc[0][8], w8, pa[0]);
                    HP_DSR(pb[(2 * KK + 11) & 3], rb_n, 2048);                       // W'[1] into W[7]'s registers
                    HP_MMA(acc[0][9], w9, pa[0]);
                    HP_DSR(pa[0], an0, 0);
                    if (!HALO && KK == 0 && dma1) issue_part(si, IC<4>{}, IC<0>{});
                    if (HALO && KK == 0 && hdo) issue_halo(m0t, y0t, h_next, IC<3>{});
                    HP_MMA(acc[1][8], w8, pa[1]);
                    HP_MMA(acc[1][9], w9, pa[1]);
                    HP_DSR(pa[1], an1, AO1);
                    HP_MMA(acc[2][8], w8, pa[2]);
                    HP_MMA(acc[2][9], w9, pa[2]);
                    HP_DSR(pa[2], an2, AO2);
                    if (HALO && KK == 0 && hdo) issue_halo(m0t, y0t, h_next, IC<4>{});
                    HP_MMA(acc[3][8], w8, pa[3]);
                    HP_MMA(acc[3][9], w9, pa[3]);
                    HP_DSR(pa[3], an3, AO3);
                    HP_DSR(pb[(2 * KK + 12) & 3], rb_n, 4096);                       // W'[2] into W[8]'s registers
                }
            };
            for (int kt = 0; kt < nk; ++kt) {
                // step 0 of tile kt: K tile 1 (kt = 0), or the tail of K tile kt + 1, into the other stage; step 1: the head of
                // K tile kt + 2 into THIS stage behind the barrier
                const bool hdo = HALO && h_next < h_nblk && kt == h_tile;
                pipe_step(IC<0>{}, kt + 1 < nk, st_c ^ 1, hdo, m0, y0);
                pipe_step(IC<1>{}, false, st_c, false, m0, y0);
                if (HALO && hdo) { ++h_next; h_tile = (9 * h_next - 10) / 2 + 1; }
                ra[0] += (unsigned)nd; ra[1] += (unsigned)nd; rb[0] += (unsigned)nd; rb[1] += (unsigned)nd;
                nd = -nd;
                st_c ^= 1;
            }
            // ... and the same on the way OUT: the epilogue's first VALU read of an accumulator must not depend on what
            // happens to sit between the last asm MFMA and it (XDL write -> VALU read, the hazard class of attention32's v_max3)
            asm volatile("s_nop 7\n\ts_nop 7" : HP_A10(0), HP_A10(1));
            asm volatile("s_nop 7\n\ts_nop 7" : HP_A10(2), HP_A10(3));
#undef HP_A10
            slot_barrier();        // every wave has finished reading the ring: the epilogue may stage into it
#undef HP_DSR
#undef HP_MMA
#undef HP_WAIT
        } else {
        if (lag) slot_barrier();
        // one K tile.  cseg / fseg (HALO_ == 2 only): this tile / the tile fetched during it belongs to the shortcut segment —
        // compile-time, and the three kinds of tile run in three loops below: with run-time tests in ONE loop hipcc peeled
        // iterations, moved accumulators between registers and spilled two of them inside the MFMA bursts (36 bytes of scratch,
        // the segment launches 4-5 % SLOWER than the per-tap kernel, profiles/r10g_conv_halo_l2_sc_probe.txt)
        auto k_tile = [&](int kt, auto cseg_tag, auto fseg_tag) {
            constexpr bool cseg = decltype(cseg_tag)::value != 0, fseg = decltype(fseg_tag)::value != 0;
            const bool more = kt + 1 < nk;
            const int si = st_c ^ 1;
            // HALO: block h_next goes into the buffer of block h_next - 2, whose last k-step (9 h_next - 10) was read in K
            // tile (9 h_next - 10) / 2 — every wave is past that read once it has passed this tile's first barrier (same
            // argument as for the ring stage below) — and is confirmed with this tile's vmcnt(0), long before its first use
            const bool hdo = HALO && !cseg && h_next < h_nblk && kt == h_tile;
            slot_barrier();                                    // 8kt
#if WIW_DMA_BURST
            // a slot that issues ANY LDS-DMA pays ~180 cycles once, further instructions ~25 each (tools/trace_probe.py)
            if (more) issue_all(si);
            if (HALO && hdo) issue_halo_all(m0, y0, h_next);
#else
            issue_slot(si, IC<0>{}, IC<fseg>{}, more, hdo, m0, y0);
#endif
            read_a(st_c, 0, kt, IC<cseg>{});
            read_b(st_c, 0, IC<0>{});
            slot_barrier();                                    // +1
#if !WIW_DMA_BURST
            issue_slot(si, IC<1>{}, IC<fseg>{}, more, hdo, m0, y0);
#endif
            mma(IC<0>{});
            slot_barrier();                                    // +2
#if !WIW_DMA_BURST
            issue_slot(si, IC<2>{}, IC<fseg>{}, more, hdo, m0, y0);
#endif
            read_b(st_c, 0, IC<1>{});
            slot_barrier();                                    // +3
#if !WIW_DMA_BURST
            issue_slot(si, IC<3>{}, IC<fseg>{}, more, hdo, m0, y0);
#endif
            mma(IC<1>{});
            slot_barrier();                                    // +4
#if !WIW_DMA_BURST
            issue_slot(si, IC<4>{}, IC<fseg>{}, more, hdo, m0, y0);
#endif
            read_a(st_c, 1, kt, IC<cseg>{});
            read_b(st_c, 1, IC<0>{});
            slot_barrier();                                    // +5
            mma(IC<0>{});
            slot_barrier();                                    // +6
            read_b(st_c, 1, IC<1>{});
            if (more) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_vmcnt<0>();                               // my share of K tile kt+1 (and anything older) has landed
            }
            slot_barrier();                                    // +7
            mma(IC<1>{});
            st_c ^= 1;
            if (HALO && hdo) { ++h_next; h_tile = (9 * h_next - 10) / 2 + 1; }
        };
        if constexpr (!HSEG) {
            for (int kt = 0; kt < nk; ++kt) k_tile(kt, IC<0>{}, IC<0>{});
        } else {
            int kt = 0;
            for (; kt < h_kc - 1; ++kt) k_tile(kt, IC<0>{}, IC<0>{});
            k_tile(kt, IC<0>{}, IC<1>{});          // the last tile of the taps fetches the first tile of the segment
            for (++kt; kt < nk; ++kt) k_tile(kt, IC<1>{}, IC<1>{});
        }
        if (!lag) slot_barrier();   // leading group: the lagging group has finished reading the ring
        }   // !PIPE

        // ---- epilogue, part 1: bias + first pass operands, then the next tile's first K tile, then the 4 passes
        constexpr int ITEMS_P = 6, ITEMS_G = 3;
        constexpr int CHr = GE ? 10 : 20, RPSr = GE ? 6 : 3;
        rederive();
        // 64 lanes = RPSr rows x CHr chunks + 4 surplus lanes; the surplus lanes (and the row slots past the 16th row of
        // a pass, below) DUPLICATE a valid lane's work — same address, same data — instead of storing to the dump page:
        // with streaming stores the dump writes were real HBM traffic (+17 % on the output stream, PMC WRITE_SIZE).
        const int lrow = lane / CHr < RPSr ? lane / CHr : RPSr - 1;
        const int lch = lane - (lane / CHr) * CHr;
        const int tile_nw = tile_n * 2 + wn;                 // index of this wave's 160-column tile
        const int ncol = (GE ? tile_nw * 80 : tile_nw * 160) + lch * 8;
        const bool lane_ok = ncol < n_valid;
        const int mw0 = m0 + wm * 64;
        float4 bvf[10];
        float4 c0, c1, rv0[2], rv1[2];
        uint4 q1[2][ITEMS_P];
        auto load_rv = [&](auto mi_tag) {
            constexpr int mi = decltype(mi_tag)::value;
            const int mr = mw0 + mi * 16;
            const int vi = (mr < p.M ? mr : p.M - 1) / p.rows_per_vec;
            const float* pa = lane_ok ? p.rowvec + (int64_t)vi * p.rowvec_ld + ncol : (const float*)zeros;
            rv0[mi & 1] = *(const float4*)pa;
            rv1[mi & 1] = *(const float4*)(lane_ok ? pa + 4 : (const float*)zeros);
        };
        auto load_q1 = [&](auto mi_tag) {
            constexpr int mi = decltype(mi_tag)::value;
#pragma unroll
            for (int k = 0; k < ITEMS_P; ++k) {
                const int rr = lrow + k * 3 < 16 ? lrow + k * 3 : 15;
                const int m = mw0 + mi * 16 + rr;
                const bool ok = lane_ok && m < p.M;
                q1[mi & 1][k] = *(const uint4*)(ok ? (const char*)(r1 + (int64_t)m * p.ldr1 + ncol) : zeros);
            }
        };
        c0 = float4{0.f, 0.f, 0.f, 0.f}; c1 = c0; rv0[0] = c0; rv0[1] = c0; rv1[0] = c0; rv1[1] = c0;
        // q1 is written and read under `if (r1)`: without an unconditional definition in THIS iteration the compiler carries
        // its 48 registers across the tile loop (and spills them around the K loop)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < ITEMS_P; ++k) q1[b][k] = uint4{0u, 0u, 0u, 0u};
        if (GE) {
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) bvf[ni] = float4{0.f, 0.f, 0.f, 0.f};   // same reason (read under `if (p.bias)`)
        }
        const bool rv_fast = p.rowvec != nullptr && (p.rows_per_vec % 16) == 0;
        // per-wave staging area: in the ring stage that held the LAST K tile (every wave has finished reading it: the
        // lagging group passed its last read before the leading group's closing barrier); the next tile's first K tile
        // goes to the other stage
        // (HALO: the free W stage holds the staging areas of waves 0..3, halo buffer 1 — the last block's, never the target of
        // the next tile's first fetch — those of waves 4..7)
        char* const stg = !HALO ? smem + (st_c ^ 1) * HSTAGE + wave * HSTG_WAVE
                                : (wave < 4 ? smem + (st_c ^ 1) * HB_BYTES + wave * HSTG_WAVE
                                            : halo0 + HALO_BUF + (wave - 4) * HSTG_WAVE);
        // plain path, step (a) of a 16-row pass: fragment layout -> 16-bit rows in LDS (lane owns row frow, columns
        // 16*ni + 4*fq .. +3).  Pass 0 is staged BEFORE the epilogue's global loads are issued: its 40 accumulator
        // registers are free by the time the 40 registers of bias / per-frame vector / residual rows arrive (with the loads
        // first the kernel peaked at 160 + 40 + addresses and spilled)
        auto stage_pass = [&](auto mi_tag) {
            constexpr int mi = decltype(mi_tag)::value;
            char* wrow = stg + frow * HSTG_ROWB + fq * 8;
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) {
                f32x4 v = acc[mi][ni];
                if (scale_acc) { v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha; }
                uint2 pk;
                pk.x = pack2bf(v[0], v[1]);
                pk.y = pack2bf(v[2], v[3]);
                *(uint2*)(wrow + ni * 32) = pk;
            }
            wave_lds_sync();
        };
        if (!GE && !SK && !F32E) stage_pass(IC<0>{});
        if (F32E) {
            // (operands are fetched per half pass in part 2)
        } else if (GE) {
            if (p.bias) {
#pragma unroll
                for (int ni = 0; ni < 10; ++ni) bvf[ni] = *(const float4*)(p.bias + tile_nw * 160 + ni * 16 + fq * 4);
            }
        } else {
            const float* zf = (const float*)zeros;
            if (p.bias) {
                const float* bp = lane_ok ? p.bias + ncol : zf;
                c0 = *(const float4*)bp; c1 = *(const float4*)(lane_ok ? bp + 4 : zf);
            }
            if (rv_fast) load_rv(IC<0>{});
            if (r1) load_q1(IC<0>{});
        }

        int q_next = q + (super ? 8 : 1);
        const int t_next = next_tile(q_next);
        if (t_next >= 0) {
            setup_loader(t_next);
            reset_loader(t_next);
            issue_all(st_c);            // stage st_c is free (it held K tile nk-2); staging uses the other one
            if constexpr (HALO) { const int m0n = (t_next / Nt) * HM; issue_halo_all(m0n, tile_y0(m0n), 0); }
        }
        pending_stores = 0;

        // ---- epilogue, part 2 (per wave, no block barrier)
        if (SK) {
            // split-K: raw fp32 partial sums straight from the fragment layout (a lane holds 4 consecutive columns of
            // one row: 16-byte stores, 64-byte runs per row) into this K range's slab; bias / residual / rounding happen
            // in splitk_reduce_kernel.  The stores are conditional, so the next tile waits with vmcnt(0).
            float* const ws = (float*)p.out + out_slab;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = mw0 + mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < 10; ++ni) {
                    const int n = tile_nw * 160 + ni * 16 + fq * 4;
                    if (m < p.M && n < p.N) {
                        float* d = ws + (int64_t)m * p.ldo + n;
                        const f32x4 v = acc[mi][ni];
                        __builtin_nontemporal_store(v[0], d); __builtin_nontemporal_store(v[1], d + 1);
                        __builtin_nontemporal_store(v[2], d + 2); __builtin_nontemporal_store(v[3], d + 3);
                    }
                }
            }
        } else if constexpr (F32E) {
            const bool out_f32 = (p.epilogue & WIW_EPI_OUT_F32) != 0;
            const bool r1_f32 = (p.epilogue & WIW_EPI_RES1_F32) != 0, r2_f32 = (p.epilogue & WIW_EPI_RES2_F32) != 0;
            auto ld4 = [&](const void* base, bool f32, int64_t off) -> float4 {
                if (f32) return *(const float4*)((const float*)base + off);
                const uint2 u = *(const uint2*)((const uint16_t*)base + off);
                const wiw_f32x2 a = unpack2(u.x), b = unpack2(u.y);
                return float4{a.x, a.y, b.x, b.y};
            };
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = mw0 + mi * 16 + frow;
                const bool m_ok = m < p.M;
                const int mc = m_ok ? m : p.M - 1;
                const float* rv = p.rowvec ? p.rowvec + (int64_t)(mc / p.rows_per_vec) * p.rowvec_ld : nullptr;
#pragma unroll
                for (int h = 0; h < 2; ++h) {       // half passes of 5 column blocks: 2 x 20 residual registers in flight
                    float4 rr1[5], rr2[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int n = tile_nw * 160 + (h * 5 + j) * 16 + fq * 4;
                        const bool ok = m_ok && n < p.N;
                        rr1[j] = (p.res1 && ok) ? ld4(p.res1, r1_f32, (int64_t)m * p.ldr1 + n) : float4{0.f, 0.f, 0.f, 0.f};
                        rr2[j] = (p.res2 && ok) ? ld4(p.res2, r2_f32, (int64_t)m * p.ldr2 + n) : float4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int n = tile_nw * 160 + (h * 5 + j) * 16 + fq * 4;
                        if (m_ok && n < p.N) {
                            f32x4 v = acc[mi][h * 5 + j];
                            if (p.bias) { const float4 b = *(const float4*)(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                            if (rv) { const float4 b = *(const float4*)(rv + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                            v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha;
                            const float4 f = rr1[j], g = rr2[j];
                            v[0] += p.beta1 * f.x + p.beta2 * g.x; v[1] += p.beta1 * f.y + p.beta2 * g.y;
                            v[2] += p.beta1 * f.z + p.beta2 * g.z; v[3] += p.beta1 * f.w + p.beta2 * g.w;
                            if (out_f32) {
                                float* d = (float*)p.out + (int64_t)m * p.ldo + n;
                                __builtin_nontemporal_store(v[0], d); __builtin_nontemporal_store(v[1], d + 1);
                                __builtin_nontemporal_store(v[2], d + 2); __builtin_nontemporal_store(v[3], d + 3);
                            } else {
                                uint32_t* d = (uint32_t*)((uint16_t*)p.out + (int64_t)m * p.ldo + n);
                                __builtin_nontemporal_store(pack2bf(v[0], v[1]), d);
                                __builtin_nontemporal_store(pack2bf(v[2], v[3]), d + 1);
                            }
                        }
                    }
                }
            }
            // (conditional stores: the next tile waits with vmcnt(0), pending_stores stays 0)
        } else {
            uint4* dump = g_dump_h + (blockIdx.x & 511) * 64 + lane;
            // plain path, step (b) of a 16-row pass: row-major, 8 consecutive columns per lane, bias / vector / residual math
            // in fp32, 16-byte streaming stores (unconditional: invalid lanes hit the dump page)
            auto pass = [&](auto mi_tag) {
                constexpr int mi = decltype(mi_tag)::value;
                constexpr int pb = mi & 1;
                constexpr int RPS = GE ? 6 : 3;
                constexpr int ITEMS = GE ? ITEMS_G : ITEMS_P;
                const int mrow0 = mw0 + mi * 16;
                if (!GE && mi < 3) {   // operands of the next pass behind this pass's math (the other buffer)
                    if (rv_fast) load_rv(IC<(mi + 1) & 3>{});
                    if (r1) load_q1(IC<(mi + 1) & 3>{});
                }
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {
                    const int rr = lrow + k * RPS < 16 ? lrow + k * RPS : 15;
                    const int m = mrow0 + rr;
                    const bool ok = lane_ok && m < p.M;
                    const uint4 sv = *(const uint4*)(stg + rr * HSTG_ROWB + lch * 16);
                    uint4 ov = sv;
                    if (!GE) {
                        float v[8], f[8];
                        unpack8(sv, v);
                        if (p.rowvec && !rv_fast) {
                            const float* rv = p.rowvec + (int64_t)((ok ? m : 0) / p.rows_per_vec) * p.rowvec_ld + (ok ? ncol : 0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += ok ? p.alpha * rv[e] : 0.f;
                        }
                        const float al = p.alpha;
                        v[0] += al * (c0.x + rv0[pb].x); v[1] += al * (c0.y + rv0[pb].y);
                        v[2] += al * (c0.z + rv0[pb].z); v[3] += al * (c0.w + rv0[pb].w);
                        v[4] += al * (c1.x + rv1[pb].x); v[5] += al * (c1.y + rv1[pb].y);
                        v[6] += al * (c1.z + rv1[pb].z); v[7] += al * (c1.w + rv1[pb].w);
                        if (r1) {
                            unpack8(q1[pb][k], f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += p.beta1 * f[e];
                        }
                        if (r2) {
                            unpack8(*(const uint4*)(ok ? (const char*)(r2 + (int64_t)m * p.ldr2 + ncol) : zeros), f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += p.beta2 * f[e];
                        }
                        ov = pack8(v);
                    }
#if WIW_ABLATE == 21
                    uint4* dst = dump;
#else
                    uint4* dst = ok ? (uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + ncol) : dump;
#endif
#if WIW_ABLATE == 20
                    *dst = ov;
#else
                    __builtin_nontemporal_store(ov.x, &dst->x); __builtin_nontemporal_store(ov.y, &dst->y);
                    __builtin_nontemporal_store(ov.z, &dst->z); __builtin_nontemporal_store(ov.w, &dst->w);
#endif
                }
                wave_lds_sync();
            };
            if constexpr (GE) {
                // GEGLU: 32 rows x 80 columns of a wave = 320 sixteen-byte chunks = exactly five 64-lane sweeps (see
                // gemm.hip): two double passes, no row written twice
                auto pass32 = [&](auto half_tag) {
                    constexpr int hh = decltype(half_tag)::value;
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2) {
                        char* wrow = stg + (m2 * 16 + frow) * HSTG_ROWB_G + fq * 8;
#if WIW_GE_PK
                        // the 20 gates of these 16 rows as two blocks of five packed pairs, breadth first (gelu_erf_pk in
                        // common.h; ten pairs at once cost 55 more spilled registers next to the 160 accumulators)
                        wiw_f32x2 ga[5], gb[5];
#pragma unroll
                        for (int ni = 0; ni < 5; ++ni) {
                            f32x4 g = acc[hh * 2 + m2][ni + 5];
                            if (p.bias) { g[0] += bvf[ni + 5].x; g[1] += bvf[ni + 5].y; g[2] += bvf[ni + 5].z; g[3] += bvf[ni + 5].w; }
                            ga[ni] = wiw_f32x2{g[0], g[1]};
                            gb[ni] = wiw_f32x2{g[2], g[3]};
                        }
                        gelu_erf_pk<5>(ga);
                        gelu_erf_pk<5>(gb);
#pragma unroll
                        for (int ni = 0; ni < 5; ++ni) {
                            f32x4 v = acc[hh * 2 + m2][ni];
                            if (p.bias) { v[0] += bvf[ni].x; v[1] += bvf[ni].y; v[2] += bvf[ni].z; v[3] += bvf[ni].w; }
                            uint2 pk;
                            pk.x = pack2bf(v[0] * ga[ni].x, v[1] * ga[ni].y);
                            pk.y = pack2bf(v[2] * gb[ni].x, v[3] * gb[ni].y);
                            *(uint2*)(wrow + ni * 32) = pk;
                        }
#else
#pragma unroll
                        for (int ni = 0; ni < 5; ++ni) {
                            f32x4 v = acc[hh * 2 + m2][ni], g = acc[hh * 2 + m2][ni + 5];
                            if (p.bias) {
                                v[0] += bvf[ni].x; v[1] += bvf[ni].y; v[2] += bvf[ni].z; v[3] += bvf[ni].w;
                                g[0] += bvf[ni + 5].x; g[1] += bvf[ni + 5].y; g[2] += bvf[ni + 5].z; g[3] += bvf[ni + 5].w;
                            }
                            uint2 pk;
                            pk.x = pack2bf(v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1]));
                            pk.y = pack2bf(v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3]));
                            *(uint2*)(wrow + ni * 32) = pk;
                        }
#endif
                    }
                    wave_lds_sync();
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int slot = k * 64 + lane;
                        const int rr = slot / 10, ch = slot - rr * 10;
                        const int m = mw0 + hh * 32 + rr;
                        const int nc = tile_nw * 80 + ch * 8;
                        const bool ok = m < p.M && nc < n_valid;
                        const uint4 ov = *(const uint4*)(stg + rr * HSTG_ROWB_G + ch * 16);
                        uint4* dst = ok ? (uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + nc) : dump;
                        __builtin_nontemporal_store(ov.x, &dst->x); __builtin_nontemporal_store(ov.y, &dst->y);
                        __builtin_nontemporal_store(ov.z, &dst->z); __builtin_nontemporal_store(ov.w, &dst->w);
                    }
                    wave_lds_sync();
                };
                pass32(IC<0>{}); pass32(IC<1>{});
                pending_stores = 10;
            } else {
                pass(IC<0>{});                                  // pass 0 was staged in part 1
                stage_pass(IC<1>{}); pass(IC<1>{});
                stage_pass(IC<2>{}); pass(IC<2>{});
                stage_pass(IC<3>{}); pass(IC<3>{});
                if (r2 == nullptr) pending_stores = 4 * ITEMS_P;
            }
        }
        t = t_next;
        q = q_next;
    }
}

template <int MODE, bool GE, bool SK, int HALO = 0, bool F32E = false, bool A1 = false>
int launch_huge(hipStream_t s, const WiwGemmArgs& a) {
    constexpr int SMEM = HALO ? HALO_SMEM : H_SMEM;
    // one-time, thread-safe setup per template instantiation: opt in to > 64 KiB of dynamic LDS, read the CU count
    static std::once_flag once;
    static bool attr_ok = false;
    static int num_cu = 256;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute((const void*)gemm_huge_kernel<MODE, GE, SK, HALO, F32E, A1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
    });
    if (!attr_ok) {
        wiw_set_error("hipFuncSetAttribute(gemm_huge) failed");
        return WIW_ELAUNCH;
    }
    const int64_t tiles = (int64_t)((a.M + HM - 1) / HM) * ((a.N + HN - 1) / HN) * (a.splitk > 1 ? a.splitk : 1);
    int64_t grid = num_cu;
    if (tiles < grid) grid = tiles;   // one tile per block (a grid that is not a multiple of 8 uses contiguous ranges)
    static const char* stg_env = getenv("WIW_GEMM_STAGGER");
    static const char* sn_env = getenv("WIW_GEMM_SN");
    const int stagger = ((stg_env ? atoi(stg_env) : 0) & 255) | ((sn_env ? atoi(sn_env) & 15 : 0) << 8);
    hipLaunchKernelGGL((gemm_huge_kernel<MODE, GE, SK, HALO, F32E, A1>), dim3((unsigned)grid), dim3(512), SMEM, s, a, stagger);
    return wiw_check_launch("wiw_gemm_bf16(huge)");
}

}  // namespace

// Shapes the 256x320 tile takes: bf16 output through the staged epilogue, full 320-wide N tiles, enough tiles to give
// (nearly) every CU one.  Everything else stays on gemm.hip's tiles.
bool wiw_gemm_huge_ok(const WiwGemmArgs& a) {
    const bool ge = (a.epilogue & WIW_EPI_GEGLU) != 0;
    const bool f32 = (a.epilogue & (WIW_EPI_OUT_F32 | WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32)) != 0;   // F32E instantiations (plain only)
    if (a.epilogue & (WIW_EPI_SILU | WIW_EPI_GELU | WIW_EPI_QUICK_GELU)) return false;
    if (f32 && (ge || a.splitk > 1 || getenv("WIW_GEMM_HUGE_NO_F32"))) return false;
    // partial last N tile (the VAE's 256 / 512-channel layers): only without GEGLU and when 320-wide tiles idle no more
    // MFMA columns than 160-wide ones would
    if (a.N % HN != 0 && (ge || ((a.N + HN - 1) / HN) * HN > ((a.N + 159) / 160) * 160)) return false;
    const int n_valid = ge ? a.n_out : a.N;
    const int al = f32 ? 4 : 8;             // fragment-layout accesses are 4 columns wide, staged rows 8
    if (n_valid % al || a.ldo % al) return false;
    if (a.res1 && a.ldr1 % al) return false;
    if (a.res2 && a.ldr2 % al) return false;
    if ((((uintptr_t)a.bias | (uintptr_t)a.rowvec) & 15) || a.rowvec_ld % 4) return false;
    // short K (< 10 K tiles) with GEGLU: the output tile's epilogue dominates and the smaller tile's finer granularity wins
    // (K = 320 GEGLU -6 %).  Plain epilogues at K = 320 were +-2 % while this kernel spilled; spill-free (round 4) the
    // 320-wide tile reads the A panel once instead of twice: M = 258 048, N = 320 138.7 -> 125.6 us with a residual,
    // 102.3 -> 88.5 without, N = 960 284 -> 250 (profiles/r10a_tile_probe.txt, r10b_tile_probe.txt)
    if (a.K < 640 && ge && !getenv("WIW_GEMM_HUGE_ANYK")) return false;
    const int64_t tiles = (int64_t)((a.M + HM - 1) / HM) * ((a.N + HN - 1) / HN);
    if (tiles < 200) return false;
    // persistent grid of one block per CU: a tile count just above a multiple of the CU count leaves the last round mostly
    // idle (M = 8064, N = 3840: 384 tiles = 1.5 rounds of 256) while the 256 x 160 tile's count fills its rounds (768 = 3.0):
    // 128 us against 107 us (tools/gemm_probe.py, profiles/r04l_tile_probe.txt).  gfx950: 256 CUs.
    const int64_t cus = 256, tiles_b = (int64_t)((a.M + HM - 1) / HM) * ((a.N + 159) / 160);
    const double fill_h = (double)tiles / (double)(((tiles + cus - 1) / cus) * cus);
    const double fill_b = (double)tiles_b / (double)(((tiles_b + cus - 1) / cus) * cus);
    if (!ge && fill_h < 0.8 && fill_b > fill_h + 0.15) return false;
    return true;
}

// WIW_K_HALO32 launches (include/wiw_svd.h): geometry and epilogue the HALO instantiation takes
bool wiw_conv_halo_shape_ok(const WiwGemmArgs& a) {
    if ((a.mode != WIW_A_CONV3X3 && a.mode != WIW_A_CONV3X3_UP) || !(a.epilogue & WIW_K_HALO32) || !(a.epilogue & WIW_W_TILED)) return false;
    if (a.mode == WIW_A_CONV3X3_UP) {   // whole output rows of ONE frame per tile, an even first row
        if ((a.Wd != 64 && a.Wd != 128) || (a.H & 1) || ((int64_t)a.H * a.Wd) % HM != 0 || a.C2 != 0 || a.C3 != 0) return false;
    }
    if (a.epilogue & (WIW_EPI_GEGLU | WIW_EPI_SILU | WIW_EPI_GELU | WIW_EPI_QUICK_GELU | WIW_EPI_LNFOLD | WIW_K_CMAJOR)) return false;
    if (a.splitk > 1) return false;
    const bool f32 = (a.epilogue & (WIW_EPI_OUT_F32 | WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32)) != 0;
    const int al = f32 ? 4 : 8;
    if (a.C1 % 64 != 0 || a.C2 % 64 != 0 || a.C3 % 64 != 0 || a.K != 9 * a.C1 + a.C2 + a.C3) return false;
    // a 256-row tile = 256 / Wd whole image rows, at most one frame edge inside it
    if ((a.Wd != 32 && a.Wd != 64 && a.Wd != 128) || a.H < 256 / a.Wd || a.M % HM != 0 || a.M % ((int64_t)a.H * a.Wd) != 0) return false;
    if (a.N % al || a.ldo % al) return false;
    if (a.res1 && a.ldr1 % al) return false;
    if (a.res2 && a.ldr2 % al) return false;
    if ((((uintptr_t)a.bias | (uintptr_t)a.rowvec) & 15) || a.rowvec_ld % 4) return false;
    if (a.N % HN != 0) return false;
    return true;
}

extern "C" int wiw_conv_halo_ok(const WiwGemmArgs* args) { return args != nullptr && wiw_conv_halo_shape_ok(*args) ? 1 : 0; }

int wiw_gemm_huge_launch(hipStream_t s, const WiwGemmArgs& a) {
    const bool ge = (a.epilogue & WIW_EPI_GEGLU) != 0;
    const bool a1 = a.mode == WIW_A_DENSE && a.C2 == 0 && a.M % HM == 0 && !getenv("WIW_GEMM_NO_A1");   // the A1 loader variant
    if (a.epilogue & WIW_K_HALO32) {
        if (!wiw_conv_halo_shape_ok(a)) {
            wiw_set_error("gemm: WIW_K_HALO32 launch outside the halo kernel's geometry (wiw_conv_halo_ok)");
            return WIW_EINVAL;
        }
        const bool f32h = (a.epilogue & (WIW_EPI_OUT_F32 | WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32)) != 0;
        if (a.mode == WIW_A_CONV3X3_UP)
            return f32h ? launch_huge<WIW_A_CONV3X3_UP, false, false, 1, true>(s, a) : launch_huge<WIW_A_CONV3X3_UP, false, false, 1>(s, a);
        if (a.C2 + a.C3 > 0)
            return f32h ? launch_huge<WIW_A_CONV3X3, false, false, 2, true>(s, a) : launch_huge<WIW_A_CONV3X3, false, false, 2>(s, a);
        return f32h ? launch_huge<WIW_A_CONV3X3, false, false, 1, true>(s, a) : launch_huge<WIW_A_CONV3X3, false, false, 1>(s, a);
    }
    if (a.splitk > 1) {   // pass 1 of a split-K launch (gemm.hip's launch() hands over the fp32 workspace as `out`)
        switch (a.mode) {
            case WIW_A_DENSE: return launch_huge<WIW_A_DENSE, false, true>(s, a);
            case WIW_A_CONV3X3: return launch_huge<WIW_A_CONV3X3, false, true>(s, a);
            case WIW_A_CONV3X3_S2: return launch_huge<WIW_A_CONV3X3_S2, false, true>(s, a);
            case WIW_A_CONV3X3_UP: return launch_huge<WIW_A_CONV3X3_UP, false, true>(s, a);
            case WIW_A_CONV3X3_S2P: return launch_huge<WIW_A_CONV3X3_S2P, false, true>(s, a);
            default: return launch_huge<WIW_A_CONV_T3, false, true>(s, a);
        }
    }
    if (a.epilogue & (WIW_EPI_OUT_F32 | WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32)) {      // fp32 residual stream (never GEGLU: huge_ok; pass 1 of a split-K launch, whose `out` is the fp32 workspace, was taken above)
        switch (a.mode) {
            case WIW_A_DENSE: return a1 ? launch_huge<WIW_A_DENSE, false, false, 0, true, true>(s, a)
                                        : launch_huge<WIW_A_DENSE, false, false, 0, true>(s, a);
            case WIW_A_CONV3X3: return launch_huge<WIW_A_CONV3X3, false, false, 0, true>(s, a);
            case WIW_A_CONV3X3_S2: return launch_huge<WIW_A_CONV3X3_S2, false, false, 0, true>(s, a);
            case WIW_A_CONV3X3_UP: return launch_huge<WIW_A_CONV3X3_UP, false, false, 0, true>(s, a);
            case WIW_A_CONV3X3_S2P: return launch_huge<WIW_A_CONV3X3_S2P, false, false, 0, true>(s, a);
            default: return launch_huge<WIW_A_CONV_T3, false, false, 0, true>(s, a);
        }
    }
    switch (a.mode) {
        case WIW_A_DENSE:
            if (a1) return ge ? launch_huge<WIW_A_DENSE, true, false, 0, false, true>(s, a) : launch_huge<WIW_A_DENSE, false, false, 0, false, true>(s, a);
            return ge ? launch_huge<WIW_A_DENSE, true, false>(s, a) : launch_huge<WIW_A_DENSE, false, false>(s, a);
        case WIW_A_CONV3X3: return launch_huge<WIW_A_CONV3X3, false, false>(s, a);
        case WIW_A_CONV3X3_S2: return launch_huge<WIW_A_CONV3X3_S2, false, false>(s, a);
        case WIW_A_CONV3X3_UP: return launch_huge<WIW_A_CONV3X3_UP, false, false>(s, a);
        case WIW_A_CONV3X3_S2P: return launch_huge<WIW_A_CONV3X3_S2P, false, false>(s, a);
        default: return launch_huge<WIW_A_CONV_T3, false, false>(s, a);
    }
}
