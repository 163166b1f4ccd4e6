// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   acc[m][n] = sum_k Agather[m][k] * W[n][k]      (both operands K-contiguous: "A . B^T")
//
// Design:
//   * PERSISTENT blocks: one block per CU (BIG) walks a contiguous range of output tiles, N fastest.
//     The LDS-DMA stream never drains at a tile boundary: the first K tiles of the NEXT output tile are
//     already in flight while the waves run the epilogue of the current one.
//   * two tile configurations of one kernel template:
//       BIG   : 256 (M) x 160 (N) x 64 (K), 512 threads = 8 waves, 3-stage LDS ring (156 KiB), two K
//               tiles of DMA in flight behind the MFMAs (counted vmcnt);
//       SMALL : 128 x 160 x 64, 4 waves, 2 stages (72 KiB, two blocks per CU) for small M.
//     Every wave owns 32 rows x ALL 160 columns = 2 x 10 fragments of v_mfma_f32_16x16x32_bf16
//     (80 fp32 accumulator VGPRs/lane), so the GEGLU value/gate pair of a column lives in ONE lane.
//     BN = 160 because every channel count of the served UNet is a multiple of 320.
//   * MFMA operands are swapped (D = W_frag x A_frag): a lane then holds 4 CONSECUTIVE output columns of
//     one row, so the epilogue packs 4 bf16 per LDS write / 16 B per global store.
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip); ONE raw
//     s_barrier per K tile; waits are counted.  The implicit-GEMM gather (3x3 / stride-2 /
//     nearest-upsample / temporal taps, zero padding) is done on the per-lane SOURCE address — padding
//     taps read a 16-byte zero buffer.
//   * LDS image: row r = 128 B (64 bf16); the 16-byte chunk c of row r is stored at chunk position
//     c ^ (r & 7) (swizzle on the DMA source chunk and again on the ds_read_b128: conflict-free).
//   * epilogue is PER WAVE and barrier-free: alpha*acc (or the GEGLU product) goes as bf16 through a
//     5 KiB per-wave LDS transpose (in the ring stage no DMA is targeting), then each lane owns 8
//     consecutive columns of a row: + alpha*(bias + per-frame vector) + residuals, 16-byte stores.
//     All global loads of the epilogue are issued before the LDS transpose so their latency overlaps it.
//   * tile order is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous range of tiles.
#include <stdlib.h>
#ifndef WIW_DMA_BURST
#define WIW_DMA_BURST 1   // 1: all DMA instructions of a K tile are issued in ONE slot (0: spread 2|2|2|1 over the four slots)
#endif
#ifndef WIW_ABLATE
#define WIW_ABLATE 0
#endif

#include <mutex>
#include <type_traits>

#include "common.h"
#ifndef WIW_GE_PK
#define WIW_GE_PK 1   // GEGLU epilogue on packed pairs, breadth first (0: value by value, the round-2 form)
#endif

// 256 x 320 tile variant (gemm_huge.hip)
bool wiw_gemm_huge_ok(const WiwGemmArgs& a);
int wiw_gemm_huge_launch(hipStream_t s, const WiwGemmArgs& a);

namespace {

constexpr int BN = 160, BK = 64;
constexpr int B_BYTES = BN * BK * 2;              // 20480
constexpr int STG_ROWB = 336;                     // bytes per staged bf16 row (160 cols + 16 B skew)
constexpr int STG_ROWB_G = 176;                   // GEGLU: 80 output cols + 16 B skew; all 32 rows of a wave are staged at once
constexpr int STG_WAVE = 32 * STG_ROWB_G;         // 5632 B per wave (plain: 16 rows x 336 B = 5376)
static_assert(16 * STG_ROWB <= STG_WAVE, "");

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

WIW_DEV void glds16(const char* g, char* l) {
#if WIW_ABLATE == 10   // no LDS-DMA at all (timing only: the MFMAs run on whatever the LDS holds)
    asm volatile("" ::"v"(g), "s"(l));
#else
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
#endif
}
// Sink for the epilogue stores / loads of lanes that own no valid output element: every epilogue VMEM
// instruction is then executed by every wave unconditionally, which makes the number of stores queued behind the
// prefetched LDS-DMA a compile-time constant (needed for the counted vmcnt waits below).
__device__ uint4 g_dump[512 * 64];

#ifdef WIW_TRACE   // debug build: block 0 records s_memtime at every slot boundary of its SECOND output tile (waves 0 and 4)
__device__ long long g_trace[2][4096];
#define WIW_TP(idx)                                                                                          \
    do {                                                                                                     \
        if (trace_on && (idx) < 4096) g_trace[wave >> 2][(idx)] = __builtin_readcyclecounter();            \
    } while (0)
#else
#define WIW_TP(idx) do { } while (0)
#endif

template <int N>
WIW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

WIW_DEV void wave_lds_sync() {   // order this wave's LDS writes before its following LDS reads (and vice versa)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// fp32-stream epilogue (F32E instantiations; WIW_EPI_OUT_F32 / _RES1_F32 / _RES2_F32 on aligned shapes): straight from the
// fragment layout — a lane holds 4 CONSECUTIVE columns of one row, the four fq lanes of a row 16 consecutive columns, so every
// access is 16 bytes per lane (8 for a 16-bit tensor) in 64-byte row runs.  fp32 accumulator + bias + vector + residuals,
// rounded ONCE (not at all for an fp32 output): no 16-bit staging in between, which is what a residual stream in fp32 is
// for.  The residual rows of both 16-row passes are requested before the first pass's math.
WIW_DEV void wiw_epilogue_f32(const WiwGemmArgs& p, const f32x4 (&acc)[2][10], int m0, int n0, int wave, int frow, int fq,
                              int64_t out_slab, bool out_f32, bool r1_f32, bool r2_f32) {
    float4 rr1[2][10], rr2[10];
    auto ld4 = [&](const void* base, bool f32, int64_t off) -> float4 {
        if (f32) return *(const float4*)((const float*)base + off);
        const uint2 u = *(const uint2*)((const uint16_t*)base + off);
        const wiw_f32x2 a = unpack2(u.x), b = unpack2(u.y);
        return float4{a.x, a.y, b.x, b.y};
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wave * 32 + mi * 16 + frow;
#pragma unroll
        for (int ni = 0; ni < 10; ++ni) {
            const int n = n0 + ni * 16 + fq * 4;
            rr1[mi][ni] = (p.res1 && m < p.M && n < p.N) ? ld4(p.res1, r1_f32, (int64_t)m * p.ldr1 + n) : float4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wave * 32 + mi * 16 + frow;
        const bool m_ok = m < p.M;
        const int mc = m_ok ? m : p.M - 1;
        const float* rv = p.rowvec ? p.rowvec + (int64_t)(mc / p.rows_per_vec) * p.rowvec_ld : nullptr;
#pragma unroll
        for (int ni = 0; ni < 10; ++ni) {
            const int n = n0 + ni * 16 + fq * 4;
            rr2[ni] = (p.res2 && m_ok && n < p.N) ? ld4(p.res2, r2_f32, (int64_t)m * p.ldr2 + n) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ni = 0; ni < 10; ++ni) {
            const int n = n0 + ni * 16 + fq * 4;
            if (m_ok && n < p.N) {
                f32x4 v = acc[mi][ni];
                if (p.bias) { const float4 b = *(const float4*)(p.bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (rv) { const float4 b = *(const float4*)(rv + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha;
                {
                    const float4 f = rr1[mi][ni];     // zeros without res1
                    v[0] += p.beta1 * f.x; v[1] += p.beta1 * f.y; v[2] += p.beta1 * f.z; v[3] += p.beta1 * f.w;
                    const float4 g = rr2[ni];
                    v[0] += p.beta2 * g.x; v[1] += p.beta2 * g.y; v[2] += p.beta2 * g.z; v[3] += p.beta2 * g.w;
                }
                if (out_f32) {
                    float* d = (float*)p.out + out_slab + (int64_t)m * p.ldo + n;
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

// LNF (WIW_EPI_LNFOLD, dense mode): the A operand is the RAW input x of a LayerNorm over K = C1 and W holds W * gamma:
//     LayerNorm(x) . W_n = rstd * (x . W'_n - mean * s_n) + t_n,   s_n = sum_k W'[n][k],  t_n = sum_k W[n][k] beta[k] + bias_n
// (exact; p.lnfold = [s | t], fp32).  Every wave owns whole rows (32 rows x all 160 columns), so it accumulates the row
// sums / sums of squares of its rows from the A fragments it feeds to the MFMAs anyway, and the fold is applied to the
// fp32 accumulators in the fragment layout BEFORE the first 16-bit rounding.  The LayerNorm pass (one read + one write
// of the activation) and the normalised tensor disappear.
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

// F32E: the instantiation that serves the fp32 residual stream (ABI 11: WIW_EPI_OUT_F32 / _RES1_F32 / _RES2_F32 on aligned
// shapes) — ONLY the vectorised fragment-layout epilogue is compiled into it, and it is compiled into no other instantiation
// (in the common one its 120 residual registers cost every launch 240-300 bytes of scratch per lane).
template <int MODE, int NW, int STAGES, bool GE, bool LNF = false, bool F32E = false>
__global__ __launch_bounds__(NW * 64, 2) void gemm_kernel(const WiwGemmArgs p, const int stagger) {
    constexpr int BM = NW * 32;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_FULL = 20 / NW;                   // full (8-row) W DMA instructions per wave
    constexpr bool B_HALF = (20 % NW) != 0;           // plus one 4-row instruction per wave (NW = 8)
    constexpr int LPT = 4 + B_FULL + (B_HALF ? 1 : 0);  // LDS-DMA instructions per wave per K tile
    constexpr int D = STAGES - 1;                     // K tiles in flight ahead of the MFMAs
    static_assert(B_FULL * NW + (B_HALF ? NW / 2 : 0) == 20, "W tile must be covered exactly");
    static_assert(NW * STG_WAVE <= STAGE_BYTES, "per-wave epilogue staging must fit in one ring stage");
    static_assert(!LNF || MODE == WIW_A_DENSE, "the LayerNorm fold needs whole rows of x in the K loop");
    constexpr int FOLD_BYTES = 2 * BN * 4;            // s | t of one 160-column tile (fp32); two parities after the ring
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;

    // ---- tile schedule.  Per-XCD L2s are private and the DMA stream is what bounds this kernel (52 KiB per
    // 5.2 MFLOP K tile: ~10 TB/s of cache-hierarchy traffic at 1 PFLOP/s), so the blocks that are resident on ONE
    // XCD at the same time work on a compact sm x sn super-tile of output tiles: every A panel is fetched into that
    // L2 once and read by sn blocks, every W panel once and read by sm blocks (blocks advance in near lock-step —
    // equal work per K tile — and 4 MiB of L2 absorbs several K tiles of skew).  Block b is assumed to run on XCD
    // b % 8 (observed dispatch order; a different placement changes speed only).
    const int Nt = (p.N + BN - 1) / BN;
    // split-K (p.splitk = S > 1): the S K-ranges of an output tile are scheduled as S extra tile rows; range ks of tile
    // row tr is schedule row ks * Mt1 + tr and writes raw fp32 sums to slab ks of the workspace (reduced, with the
    // whole epilogue, by splitk_reduce_kernel).  Used where M alone cannot fill the chip (M = 4032 at the 1280 level).
    const int S = p.splitk > 1 ? p.splitk : 1;
    const int Mt1 = (p.M + BM - 1) / BM;
    const int Mt = Mt1 * S;
    const int total = Mt * Nt;
    const int nb = gridDim.x;
    const bool super = (nb & 7) == 0 && nb >= 64;      // otherwise: small grid, contiguous ranges
    // sn: power of two <= 8 (divides nb / 8) with the fewest padded (idle) tile slots; ties go to the wider super-tile
    // (Nt = 6 with sn = 4 would leave a quarter of the blocks without work in every second super-tile column)
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
    const int sm = bpx / sn;
    const int SNt = (Nt + sn - 1) / sn, SMt = (Mt + sm - 1) / sm;
    const int n_super = SNt * SMt;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    int q = xcd, t_end = 0;
    // next output tile of this block at or after schedule position `pos` (-1: done); advances pos
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
    // De-phase the blocks: every block has the same work per output tile, so without this all CUs run their main
    // loops together and then write their tiles together; the epilogue's store burst (80 KB per CU) then drains at
    // the HBM write rate while every MFMA pipe waits (stores retire in order ahead of the next tile's LDS-DMA).
    if (stagger > 0) {
        const int phase = ((blockIdx.x >> 3) + 2 * (blockIdx.x >> 8)) & 3;
        for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(16);   // 1024 cycles per iteration
    }

    // ---- loader state (per-thread source rows of the tile being fetched)
    const int rsub = lane >> 3;                       // row inside the 8-row group of one DMA instruction
    const int chunk = (lane & 7) ^ (rsub & 7);        // logical 16-B chunk fetched into position lane&7
    const char* const Ab = (const char*)p.A;
    const char* const A2b = (const char*)p.A2;
    const char* const A3b = (const char*)p.A3;
    const char* const zeros = (const char*)p.zeros;
    const int HW = p.H * p.Wd;
    const int Ctot = MODE == WIW_A_DENSE ? p.C1 + p.C2 : p.C1;   // channels per tap (the shortcut segment is tap 9)
    const int nk = p.K / BK / S;       // K tiles per output tile (of one split range)
    const int nk_w = p.K / BK;         // K tiles per 8-row block of a tiled W

    int a_m[4];
    bool a_ok[4];
    int a_fb[4], a_y[4], a_x[4];   // conv: input-frame base row, output y / x  (temporal: a_y = t)
    const char* w_row[B_FULL + 1];
    // WIW_W_TILED: W pre-tiled by the host into 1-KiB blocks [n / 8][k tile][8 rows x 128 B, chunks pre-swizzled]: one
    // DMA instruction then reads ONE contiguous KiB (63 B/clk/CU) instead of 8 row segments K*2 bytes apart (25 B/clk/CU,
    // tools/ubench/lds_fill.hip) — the LDS image is byte-identical
    const bool w_tiled = (p.epilogue & WIW_W_TILED) != 0;
    const int64_t w_kstep = w_tiled ? 1024 : BK * 2;
    int ld_tap = 0, ld_cc = 0, ld_kt = 0;
    // K walk of the implicit GEMM.  Tap-major (k = tap * Ctot + c) or, with WIW_K_CMAJOR, channel-block-major
    // (k = ((c / 64) * taps + tap) * 64 + c % 64): the taps of one 64-channel block are consecutive K tiles and re-read the
    // same activation window while it is still in the XCD's L2 (include/wiw_svd.h).  The shortcut segment (tap 9) is last.
    constexpr int NTAPS = MODE == WIW_A_CONV_T3 ? 3 : 9;
    const bool cmajor = MODE != WIW_A_DENSE && (p.epilogue & WIW_K_CMAJOR) != 0;

    auto setup_loader = [&](int tile) {
        const int m0 = ((tile / Nt) % Mt1) * BM, n0 = (tile % Nt) * BN;
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
        const int w_last = ((p.N + 7) >> 3) - 1;   // tiled W: last 8-row block (blocks past N re-read it: computed, never stored)
#pragma unroll
        for (int i = 0; i < B_FULL; ++i) {
            if (w_tiled) {
                int blk = (n0 >> 3) + wave * B_FULL + i;
                blk = blk < w_last ? blk : w_last;
                w_row[i] = (const char*)p.W + (int64_t)blk * nk_w * 1024 + lane * 16;
            } else {
                int n = n0 + (wave * B_FULL + i) * 8 + rsub;
                n = n < p.N ? n : p.N - 1;   // clamped rows are computed but never stored
                w_row[i] = (const char*)p.W + ((int64_t)n * p.K + chunk * 8) * 2;
            }
        }
        if (w_tiled) {   // NW = 8: rows 128..159 = blocks 16..19, half a block (4 rows = 512 contiguous bytes) per wave
            int blk = (n0 >> 3) + B_FULL * NW + (wave >> 1);
            blk = blk < w_last ? blk : w_last;
            w_row[B_FULL] = (const char*)p.W + (int64_t)blk * nk_w * 1024 + (wave & 1) * 512 + (lane & 31) * 16;
        } else {   // NW = 8: rows 128..159 of the W tile are fetched 4 rows per wave by lanes 0..31
            const int r_h = B_FULL * NW * 8 + wave * 4 + (rsub & 3);   // r_h & 7 != rsub for odd waves
            const int chunk_h = (lane & 7) ^ (r_h & 7);
            int n = n0 + r_h;
            n = n < p.N ? n : p.N - 1;
            w_row[B_FULL] = (const char*)p.W + ((int64_t)n * p.K + chunk_h * 8) * 2;
        }
    };

    // K position of the loader at the start of `tile`: K tile ks * nk of its split range
    auto reset_loader = [&](int tile) {
        ld_kt = S > 1 ? ((tile / Nt) / Mt1) * nk : 0;
        const int k0 = ld_kt * BK;
        if (MODE == WIW_A_DENSE) { ld_tap = 0; ld_cc = k0; }
        else {
            if (cmajor) {
                const int kt0 = k0 / BK, nblk = Ctot / BK;
                if (kt0 < NTAPS * nblk) { ld_tap = kt0 % NTAPS; ld_cc = (kt0 / NTAPS) * BK; }
                else { ld_tap = 9; ld_cc = (kt0 - NTAPS * nblk) * BK; }
                return;
            }
            int tp = k0 / Ctot;
            if (MODE == WIW_A_CONV3X3 && tp > 9) tp = 9;     // inside the fused-shortcut segment (longer than one tap)
            ld_tap = tp; ld_cc = k0 - tp * Ctot;
        }
    };

    // address of the 16 bytes this lane fetches for A row i of K tile (tap, cc)
    auto a_src = [&](int i, int tap, int cc) -> const char* {
        if (MODE == WIW_A_DENSE) {
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
                constexpr int o = (MODE == WIW_A_CONV3X3_S2P) ? 1 : 0;   // S2P: padding on the bottom / right only
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

    // issue the DMA of the loader's next K tile into ring stage `stage`
    auto issue_next = [&](int stage) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4 * 1024;
        char* sB = smem + stage * STAGE_BYTES + A_BYTES + wave * B_FULL * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src(i, ld_tap, ld_cc), sA + i * 1024);
#pragma unroll
        for (int i = 0; i < B_FULL; ++i) glds16(w_row[i] + (int64_t)ld_kt * w_kstep, sB + i * 1024);
        if (B_HALF) {
            char* sH = smem + stage * STAGE_BYTES + A_BYTES + B_FULL * NW * 1024 + wave * 512;
            if (lane < 32) glds16(w_row[B_FULL] + (int64_t)ld_kt * w_kstep, sH);
        }
        ++ld_kt;
        { const WiwKCur kc = wiw_advance_k<MODE, BK>(WiwKCur{ld_tap, ld_cc}, Ctot, cmajor); ld_tap = kc.tap; ld_cc = kc.cc; }
    };

    // BIG: the 7 DMA instructions of a K tile are spread 2|2|2|1 over the four slots of the previous tile's
    // iteration: an LDS-DMA instruction costs the issuing wave ~100 cycles, and a slot that carries all 7 is three
    // times longer than the MFMA slot of the partner wave it is supposed to overlap.
    auto issue_part = [&](int stage, auto part_tag) {
        constexpr int part = decltype(part_tag)::value;
        char* sA = smem + stage * STAGE_BYTES + wave * 4 * 1024;
        char* sB = smem + stage * STAGE_BYTES + A_BYTES + wave * B_FULL * 1024;
        if (part < 2) {
            glds16(a_src(2 * part, ld_tap, ld_cc), sA + (2 * part) * 1024);
            glds16(a_src(2 * part + 1, ld_tap, ld_cc), sA + (2 * part + 1) * 1024);
        } else if (part == 2) {
#pragma unroll
            for (int i = 0; i < B_FULL; ++i) glds16(w_row[i] + (int64_t)ld_kt * w_kstep, sB + i * 1024);
        } else {
            if (B_HALF) {
                char* sH = smem + stage * STAGE_BYTES + A_BYTES + B_FULL * NW * 1024 + wave * 512;
                if (lane < 32) glds16(w_row[B_FULL] + (int64_t)ld_kt * w_kstep, sH);
            }
            ++ld_kt;
            { const WiwKCur kc = wiw_advance_k<MODE, BK>(WiwKCur{ld_tap, ld_cc}, Ctot, cmajor); ld_tap = kc.tap; ld_cc = kc.cc; }
        }
    };

    f32x4 acc[2][10];
    // Fragment registers of ONE 32-deep k-step: filled in a wave's "read" slot, consumed in its "MFMA" slot.
    // Every wave runs  R(kk=0) | M | R(kk=1) | M  per K tile with a barrier between slots; the second wave group
    // (waves NW/2..NW-1, one per SIMD next to a wave of the first group) runs ONE SLOT BEHIND, so on every SIMD one
    // wave streams ds_read_b128 (and issues the DMA) while the other issues 20 MFMAs back to back.
    bf16x8 fa[2], fb[10];
    auto read_frags = [&](int stage, int kk) {
#if WIW_ABLATE == 2
        return;
#endif
        const int sw = ((kk * 4 + fq) ^ (lane & 7)) << 4;
        const char* sA = smem + stage * STAGE_BYTES + (wave * 32 + frow) * 128 + sw;
        const char* sB = smem + stage * STAGE_BYTES + A_BYTES + frow * 128 + sw;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const bf16x8*)(sA + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < 10; ++ni) fb[ni] = *(const bf16x8*)(sB + ni * 2048);
    };
    float sum1[2] = {0.f, 0.f}, sum2[2] = {0.f, 0.f};   // LNF: per-lane partial row sums of x and x^2 (rows 16*mi + frow)
    char* const fold_scr = smem + STAGES * STAGE_BYTES;   // LNF: [parity][s(160) | t(160)]
    int fold_par = 0;
    // LNF: waves 0 / 1 fetch s / t of tile `tile`'s 160 columns (40 lanes x 16 B each) into parity `par`
    auto issue_fold = [&](int tile, int par) {
        if (LNF && wave < 2 && lane < 40) {
            const int n0f = (tile % Nt) * BN;
            glds16((const char*)(p.lnfold + (int64_t)wave * p.N + n0f + lane * 4), fold_scr + par * FOLD_BYTES + wave * (BN * 4));
        }
    };
    auto mma = [&]() {
#if WIW_ABLATE == 1
        asm volatile("" :: "v"(fa[0]), "v"(fb[0]));
        return;
#endif
#if WIW_ABLATE != 3
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int ni = 0; ni < 10; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)   // swapped operands: lane gets n = 16*ni + 4*fq + r, m = 16*mi + frow
                acc[mi][ni] = WIW_MFMA(fb[ni], fa[mi], acc[mi][ni]);
#if WIW_ABLATE != 3
        __builtin_amdgcn_s_setprio(0);
#endif
        if (LNF) {   // row statistics from the A fragments just consumed (VALU, behind the MFMAs in flight)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                union { bf16x8 v; uint32_t u[4]; } x;
                x.v = fa[mi];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const wiw_f32x2 lh = unpack2(x.u[j]);
                    sum1[mi] += lh.x + lh.y;
                    sum2[mi] = __builtin_fmaf(lh.x, lh.x, __builtin_fmaf(lh.y, lh.y, sum2[mi]));
                }
            }
        }
    };
    auto slot_barrier = [&]() {   // close a slot: this wave's LDS reads are retired, then rendezvous
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto slot_barrier_i = [&]() {   // intra-tile slot boundary
#if WIW_ABLATE != 4
        slot_barrier();
#endif
    };
    const bool lag = wave >= NW / 2;   // wave-uniform: second wave group

    constexpr bool geglu = GE;   // compile-time: keeps the GEGLU-only / residual-only epilogue registers apart
    const bool do_silu = (p.epilogue & WIW_EPI_SILU) != 0;
    const int act = p.epilogue & (WIW_EPI_GELU | WIW_EPI_QUICK_GELU);   // CLIP MLP activations (direct path only)
    const bool scale_acc = p.alpha != 1.0f;   // wave-uniform: most GEMMs skip the alpha multiply
    const bool out_f32 = (p.epilogue & WIW_EPI_OUT_F32) != 0;
    const int n_valid = geglu ? p.n_out : p.N;
    const uint16_t* r1 = (const uint16_t*)p.res1;
    const uint16_t* r2 = (const uint16_t*)p.res2;
    // fp32 residual stream (ABI 11): res1 / res2 / out in fp32 -> the vectorised fragment-layout epilogue below
    const bool r1_f32 = (p.epilogue & WIW_EPI_RES1_F32) != 0, r2_f32 = (p.epilogue & WIW_EPI_RES2_F32) != 0;
    // staged (fast) epilogue: bf16 output on 16-byte aligned rows; everything else takes the direct path
    const bool staged = !F32E && !out_f32 && !r1_f32 && !r2_f32 && !do_silu && act == 0 && (n_valid % 8 == 0) && (p.ldo % 8 == 0) &&
                        (r1 == nullptr || p.ldr1 % 8 == 0) && (r2 == nullptr || p.ldr2 % 8 == 0) &&
                        ((((uintptr_t)p.bias | (uintptr_t)p.rowvec) & 15) == 0) && (p.rowvec_ld % 4 == 0);

    // ---- prologue: first D K tiles of the first output tile
    setup_loader(t);
    reset_loader(t);
    issue_fold(t, 0);
    int st_c = 0;   // ring stage holding the K tile the MFMAs consume next
    int pending_stores = 0;   // epilogue stores of the previous tile queued behind the prefetched DMA (0 / 5 / 12)
#ifdef WIW_TRACE
    int tiles_done = 0;
#endif
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < nk) issue_next(j);

    while (t >= 0) {
#ifdef WIW_TRACE
        const bool trace_on = blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && tiles_done == 1;
#endif
        const int tile_n = t % Nt;
        const int m0 = ((t / Nt) % Mt1) * BM, n0 = tile_n * BN;
        const int64_t out_slab = S > 1 ? (int64_t)((t / Nt) / Mt1) * p.M * p.ldo : 0;   // fp32 workspace slab of this K range
        setup_loader(t);   // recomputed (not kept live across the previous epilogue: VGPR budget); counters ld_* persist
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        sum1[0] = sum1[1] = sum2[0] = sum2[1] = 0.f;

        // ---- main loop.  Local slots of K tile kt:  [4kt] issue DMA(kt+D), R(kt,0)  [4kt+1] M  [4kt+2] R(kt,1),
        // wait(tile kt+1)  [4kt+3] M.   The lagging group executes one extra barrier before its first slot and the
        // leading group one after its last, so a lagging wave's local barrier b is the leading waves' barrier b+1:
        //   * DMA into the stage of tile kt-1 is issued after local barrier 4kt; by then the other group has passed
        //     its barrier 4kt-1, i.e. finished R(kt-1,1) — the last read of that stage;
        //   * tile j is first read in local slot 4j of the LEADING group = barrier 4j-1 of the lagging group, so
        //     every wave confirms "my DMA portion of tile j landed" before its local barrier 4j-1 (end of slot 4j-2).
        // vmcnt retires in issue order (loads and stores share the queue on gfx9-class parts): "tile j landed" ==
        // "at most <ops issued after it> outstanding" = the DMA of tile j+1 (BIG) plus, for the first D tiles of a
        // non-first output tile, the previous epilogue's stores (compile-time count: they are unconditional) — the
        // stores drain behind the MFMAs instead of stalling the wave at the tile boundary.
        // wait until this wave's DMA portion of K tile j has landed; `n_dma` = DMA instructions issued after it
        auto wait_tile = [&](int j, auto ndma_tag) {
            constexpr int n_dma = decltype(ndma_tag)::value;
            const bool dma_younger = (D == 2) && (j + 1 < nk);
            const int st_younger = (j < D) ? pending_stores : 0;   // wave-uniform: 0, 5 (GEGLU) or 12
            if (dma_younger) {
                if (st_younger == 12) wait_vmcnt<n_dma + 12>();
                else if (st_younger == 5) wait_vmcnt<n_dma + 5>();
                else wait_vmcnt<n_dma>();
            } else {
                if (st_younger == 12) wait_vmcnt<12>();
                else if (st_younger == 5) wait_vmcnt<5>();
                else wait_vmcnt<0>();
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        using ILPT = std::integral_constant<int, LPT>; using I6 [[maybe_unused]] = std::integral_constant<int, LPT - 1>;
        wait_tile(0, ILPT{});
        if (NW == 8) {
            WIW_TP(0);
            if (lag) slot_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                slot_barrier();                                  // local barrier 4kt
                WIW_TP(1 + 8 * kt);
                const bool more = kt + D < nk;
                int si = st_c + D;
                si = si >= STAGES ? si - STAGES : si;
#if WIW_DMA_BURST
                // A slot that issues ANY LDS-DMA pays ~180 cycles once, every further instruction of the same slot ~25
                // (tools/trace_probe.py): the 7 instructions of K tile kt+2 go out together here instead of 2|2|2|1
                if (more) { issue_part(si, I0{}); issue_part(si, I1{}); issue_part(si, I2{}); issue_part(si, I3{}); }
#else
                if (more) issue_part(si, I0{});
#endif
                read_frags(st_c, 0);
                WIW_TP(2 + 8 * kt);
                slot_barrier_i();                                // 4kt+1
                WIW_TP(3 + 8 * kt);
#if !WIW_DMA_BURST
                if (more) issue_part(si, I1{});
#endif
                mma();
                WIW_TP(4 + 8 * kt);
                slot_barrier_i();                                // 4kt+2
                WIW_TP(5 + 8 * kt);
#if !WIW_DMA_BURST
                if (more) issue_part(si, I2{});
#endif
                read_frags(st_c, 1);
                if (kt + 1 < nk) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if WIW_DMA_BURST
                    wait_tile(kt + 1, ILPT{});                   // all 7 DMA instructions of tile kt+2 are younger
#else
                    wait_tile(kt + 1, I6{});                     // 6 of tile kt+2's 7 DMA instructions are younger
#endif
                }
                WIW_TP(6 + 8 * kt);
                slot_barrier_i();                                // 4kt+3
                WIW_TP(7 + 8 * kt);
#if !WIW_DMA_BURST
                if (more) issue_part(si, I3{});
#endif
                mma();
                WIW_TP(8 + 8 * kt);
                st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
            }
            if (!lag) slot_barrier();   // leading group: the lagging group has finished reading the ring
            WIW_TP(1 + 8 * nk);
        } else {
            // SMALL (4 waves, one per SIMD, two blocks per CU): the co-resident block provides the overlap; one
            // barrier per K tile, both k-steps back to back
            for (int kt = 0; kt < nk; ++kt) {
                slot_barrier();
                if (kt + D < nk) {
                    int si = st_c + D;
                    si = si >= STAGES ? si - STAGES : si;
                    issue_next(si);
                }
                read_frags(st_c, 0);
                mma();
                read_frags(st_c, 1);
                mma();
                if (kt + 1 < nk) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wait_tile(kt + 1, ILPT{});
                }
                st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
            }
            slot_barrier();             // every wave is done reading the ring
        }

        // ---- epilogue, part 1: issue EVERY global load of the epilogue (bias, per-frame vector, residual) now,
        // BEFORE the next tile's DMA: waiting for them later then never has to wait for anything younger.
        // Lane -> output mapping of the row-major store phase: a lane keeps ONE 16-byte column chunk (lch) and
        // sweeps rows lrow, lrow + RPS, ...; lanes / rows without a valid element read zeros and write g_dump.
        const bool fast_st = staged && r2 == nullptr;   // counted-wait path (res2 would need loads behind the stores)
        constexpr int ITEMS_P = 6, ITEMS_G = 3;         // row sweeps per 16-row pass: plain (RPS 3) / GEGLU (RPS 6)
        constexpr int CHr = GE ? 10 : 20, RPSr = GE ? 6 : 3;
        // 64 lanes = RPSr rows x CHr chunks + 4 surplus lanes; the surplus lanes (and the row slots past the 16th row of
        // a pass, below) DUPLICATE a valid lane's work — same address, same data — instead of storing to the dump page:
        // with streaming stores the dump writes were real HBM traffic (+17 % on the output stream, PMC WRITE_SIZE).
        const int lrow = lane / CHr < RPSr ? lane / CHr : RPSr - 1;
        const int lch = lane - (lane / CHr) * CHr;
        const int ncol = (geglu ? tile_n * 80 : n0) + lch * 8;
        const bool lane_ok = ncol < n_valid;
        const int mw0 = m0 + wave * 32;                  // first row of this wave's 32-row band
        float4 bvf[10];                                   // GEGLU: bias in fragment layout (value | gate)
        float4 c0, c1, rv0[2], rv1[2];                    // plain: bias and the per-frame vector row of each 16-row pass
        uint4 q1[2][ITEMS_P];
        auto load_rv = [&](auto mi_tag) {
            constexpr int mi = decltype(mi_tag)::value;
            const int mr = mw0 + mi * 16;
            const int vi = (mr < p.M ? mr : p.M - 1) / p.rows_per_vec;
            const float* pa = lane_ok ? p.rowvec + (int64_t)vi * p.rowvec_ld + ncol : (const float*)zeros;
            rv0[mi] = *(const float4*)pa;
            rv1[mi] = *(const float4*)(lane_ok ? pa + 4 : (const float*)zeros);
        };
        auto load_q1 = [&](auto mi_tag) {   // residual rows of one 16-row pass (invalid lanes read the zero page)
            constexpr int mi = decltype(mi_tag)::value;
#pragma unroll
            for (int k = 0; k < ITEMS_P; ++k) {
                const int rr = lrow + k * 3 < 16 ? lrow + k * 3 : 15;
                const int m = mw0 + mi * 16 + rr;
                const bool ok = lane_ok && m < p.M;
                q1[mi][k] = *(const uint4*)(ok ? (const char*)(r1 + (int64_t)m * p.ldr1 + ncol) : zeros);
            }
        };
        c0 = float4{0.f, 0.f, 0.f, 0.f}; c1 = c0; rv0[0] = c0; rv0[1] = c0; rv1[0] = c0; rv1[1] = c0;
        // per-frame vector: when rows_per_vec is a multiple of 16 every 16-row pass sees ONE vector row
        const bool rv_fast = p.rowvec != nullptr && (p.rows_per_vec % 16) == 0;
        if (staged) {
            if (geglu) {
                if (p.bias) {
#pragma unroll
                    for (int ni = 0; ni < 10; ++ni) bvf[ni] = *(const float4*)(p.bias + n0 + ni * 16 + fq * 4);
                }
            } else {
                const float* zf = (const float*)zeros;
                if (p.bias) {
                    const float* bp = lane_ok ? p.bias + ncol : zf;
                    c0 = *(const float4*)bp; c1 = *(const float4*)(lane_ok ? bp + 4 : zf);
                }
                if (rv_fast) load_rv(std::integral_constant<int, 0>{});
                if (r1) load_q1(std::integral_constant<int, 0>{});
            }
        }

        // ---- keep the DMA stream going: first D K tiles of the NEXT output tile -> stages st_c, st_c+1.
        // Stage (st_c + D) % STAGES is not targeted until the next tile's first barrier: epilogue staging.
        int q_next = q + (super ? 8 : 1);
        const int t_next = next_tile(q_next);
        if (t_next >= 0) {
            setup_loader(t_next);
            reset_loader(t_next);
            issue_fold(t_next, fold_par ^ 1);   // the other parity was last read in the previous tile's epilogue
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (j < nk) {
                    int sj = st_c + j;
                    sj = sj >= STAGES ? sj - STAGES : sj;
                    issue_next(sj);
                }
            }
        }
        int st_e = st_c + D;
        st_e = st_e >= STAGES ? st_e - STAGES : st_e;
        pending_stores = 0;

        // LNF: finish the row statistics (the four lanes fq = 0..3 of a row hold partial sums) and the fold factors
        float ln_rs[2] = {1.f, 1.f}, ln_nrm[2] = {0.f, 0.f};
        const char* const fscr = fold_scr + fold_par * FOLD_BYTES;
        if (LNF) {
            const float inv_k = 1.0f / (float)p.K;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const float s1 = xor32_sum(xor16_sum(sum1[mi])), s2 = xor32_sum(xor16_sum(sum2[mi]));
                const float mean = s1 * inv_k;
                const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_k), 0.f);
                ln_rs[mi] = rsqrtf(var + p.ln_eps);
                ln_nrm[mi] = -ln_rs[mi] * mean;
            }
            fold_par ^= 1;
        }
        // x = rstd * acc + (t - rstd * mean * s) for the 4 columns 16*ni + 4*fq .. +3 of fragment ni
        auto ln_fold4 = [&](f32x4& v, int ni, int mi) {
            const float4 sv = *(const float4*)(fscr + (ni * 16 + fq * 4) * 4);
            const float4 tv = *(const float4*)(fscr + BN * 4 + (ni * 16 + fq * 4) * 4);
            v[0] = __builtin_fmaf(ln_rs[mi], v[0], __builtin_fmaf(ln_nrm[mi], sv.x, tv.x));
            v[1] = __builtin_fmaf(ln_rs[mi], v[1], __builtin_fmaf(ln_nrm[mi], sv.y, tv.y));
            v[2] = __builtin_fmaf(ln_rs[mi], v[2], __builtin_fmaf(ln_nrm[mi], sv.z, tv.z));
            v[3] = __builtin_fmaf(ln_rs[mi], v[3], __builtin_fmaf(ln_nrm[mi], sv.w, tv.w));
        };

        // ---- epilogue, part 2 (per wave, no block barrier)
        if constexpr (F32E) {
            wiw_epilogue_f32(p, acc, m0, n0, wave, frow, fq, out_slab, out_f32, r1_f32, r2_f32);
        } else if (staged) {
            char* stg = smem + st_e * STAGE_BYTES + wave * STG_WAVE;
            uint4* dump = g_dump + (blockIdx.x & 511) * 64 + lane;
            auto half_pass = [&](auto ge_tag, auto mi_tag) {
                constexpr int mi = decltype(mi_tag)::value;    // compile-time: acc[mi] must stay in registers
                static_assert(decltype(ge_tag)::value == GE, "");
                constexpr int RPS = GE ? 6 : 3;                // rows per sweep
                constexpr int ITEMS = GE ? ITEMS_G : ITEMS_P;
                const int mrow0 = mw0 + mi * 16;               // first of the 16 rows handled in this pass
                // (a) fragment layout -> bf16 rows in LDS: lane owns row frow, columns 16*ni + 4*fq .. +3
                char* wrow = stg + frow * STG_ROWB + fq * 8;
                if (GE) {
#pragma unroll
                    for (int ni = 0; ni < 5; ++ni) {
                        f32x4 v = acc[mi][ni], g = acc[mi][ni + 5];
                        if (p.bias) {
                            v[0] += bvf[ni].x; v[1] += bvf[ni].y; v[2] += bvf[ni].z; v[3] += bvf[ni].w;
                            g[0] += bvf[ni + 5].x; g[1] += bvf[ni + 5].y; g[2] += bvf[ni + 5].z; g[3] += bvf[ni + 5].w;
                        }
                        // (a packed-fp32 polynomial erf without v_exp / v_rcp was measured 2 % SLOWER here)
                        const wiw_f32x2 h01 = {v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1])};
                        const wiw_f32x2 h23 = {v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3])};
                        uint2 pk;
                        pk.x = pack2bf(h01.x, h01.y);
                        pk.y = pack2bf(h23.x, h23.y);
                        *(uint2*)(wrow + ni * 32) = pk;
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 10; ++ni) {
                        f32x4 v = acc[mi][ni];
                        if (LNF) ln_fold4(v, ni, mi);
                        if (scale_acc) { v[0] *= p.alpha; v[1] *= p.alpha; v[2] *= p.alpha; v[3] *= p.alpha; }
                        uint2 pk;
                        pk.x = pack2bf(v[0], v[1]);
                        pk.y = pack2bf(v[2], v[3]);
                        *(uint2*)(wrow + ni * 32) = pk;
                    }
                }
                wave_lds_sync();
                if (!GE && mi == 0) {   // acc[0]'s registers are free now: fetch pass 1's operands behind pass 0's math
                    if (rv_fast) load_rv(std::integral_constant<int, 1>{});
                    if (r1) load_q1(std::integral_constant<int, 1>{});
                }
                // (b) row-major: 8 consecutive columns per lane, residual / bias math in fp32, 16-byte stores.
                //     The store is UNCONDITIONAL (invalid lanes hit g_dump) so that exactly ITEMS stores are issued.
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {
                    const int rr = lrow + k * RPS < 16 ? lrow + k * RPS : 15;
                    const int m = mrow0 + rr;
                    const bool ok = lane_ok && m < p.M;
                    const uint4 sv = *(const uint4*)(stg + rr * STG_ROWB + lch * 16);   // rr <= 17 stays inside the stage
                    uint4 ov = sv;
                    if (!GE) {
                        float v[8], f[8];
                        unpack8(sv, v);
                        if (p.rowvec && !rv_fast) {      // odd rows_per_vec (tiny shapes): vector row per item
                            const float* rv = p.rowvec + (int64_t)((ok ? m : 0) / p.rows_per_vec) * p.rowvec_ld + (ok ? ncol : 0);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += ok ? p.alpha * rv[e] : 0.f;
                        }
                        const float al = p.alpha;
                        v[0] += al * (c0.x + rv0[mi].x); v[1] += al * (c0.y + rv0[mi].y);
                        v[2] += al * (c0.z + rv0[mi].z); v[3] += al * (c0.w + rv0[mi].w);
                        v[4] += al * (c1.x + rv1[mi].x); v[5] += al * (c1.y + rv1[mi].y);
                        v[6] += al * (c1.z + rv1[mi].z); v[7] += al * (c1.w + rv1[mi].w);
                        if (r1) {
                            unpack8(q1[mi][k], f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += p.beta1 * f[e];
                        }
                        if (r2) {   // only the AlphaBlender GEMM: loaded late, this path drains at the next tile
                            unpack8(*(const uint4*)(ok ? (const char*)(r2 + (int64_t)m * p.ldr2 + ncol) : zeros), f);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += p.beta2 * f[e];
                        }
                        ov = pack8(v);
                    }
#if WIW_ABLATE == 5
                    uint4* dst = dump;
#elif WIW_ABLATE == 7   // same bytes, but every store instruction writes 1 KiB contiguous (wrong layout: timing only)
                    const int64_t lin = ((((int64_t)t * NW + wave) * 12 + (mi * 6 + k)) * 1024 + lane * 16) %
                                        ((int64_t)p.M * p.ldo * 2 - 1024);
                    uint4* dst = (uint4*)((char*)p.out + (lin & ~15ll));
#else
                    uint4* dst = ok ? (uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + ncol) : dump;
#endif
#if WIW_ABLATE == 9
                    *dst = ov;
#else
                    // streaming (nt) store: the output tile is never re-read by this kernel, and a plain store
                    // allocates it in the XCD's 4 MiB L2 where it evicts the A / W panels the super-tile schedule
                    // keeps there (measured -17...-26 % on the K <= 1280 projections at L0)
                    __builtin_nontemporal_store(ov.x, &dst->x); __builtin_nontemporal_store(ov.y, &dst->y);
                    __builtin_nontemporal_store(ov.z, &dst->z); __builtin_nontemporal_store(ov.w, &dst->w);
#endif
                }
                wave_lds_sync();
            };
#if WIW_ABLATE == 6
            if (p.M < 0) {
#endif
            if constexpr (GE) {
                // GEGLU: the output tile of a wave is 32 rows x 80 columns = 320 sixteen-byte chunks = EXACTLY five
                // 64-lane sweeps (the 16-row passes above would take 2 x 3 sweeps of 6 rows and write two rows twice per
                // pass: +12.5 % on the output stream, PMC WRITE_SIZE 743 MB for 660 MB algorithmic at L0).  No per-column
                // math is left for the row-major phase (bias and GELU are applied in the fragment layout), so a lane may
                // own a different column chunk in every sweep.
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    char* wrow = stg + (mi * 16 + frow) * STG_ROWB_G + fq * 8;
#if WIW_GE_PK
                    // the 20 gates of these 16 rows as ten packed pairs, breadth first (gelu_erf_pk in common.h)
                    wiw_f32x2 gg[10];
#pragma unroll
                    for (int ni = 0; ni < 5; ++ni) {
                        f32x4 g = acc[mi][ni + 5];
                        if (LNF) ln_fold4(g, ni + 5, mi);          // t carries the bias
                        else if (p.bias) { g[0] += bvf[ni + 5].x; g[1] += bvf[ni + 5].y; g[2] += bvf[ni + 5].z; g[3] += bvf[ni + 5].w; }
                        gg[2 * ni] = wiw_f32x2{g[0], g[1]};
                        gg[2 * ni + 1] = wiw_f32x2{g[2], g[3]};
                    }
                    gelu_erf_pk<10>(gg);
#pragma unroll
                    for (int ni = 0; ni < 5; ++ni) {
                        f32x4 v = acc[mi][ni];
                        if (LNF) ln_fold4(v, ni, mi);
                        else if (p.bias) { v[0] += bvf[ni].x; v[1] += bvf[ni].y; v[2] += bvf[ni].z; v[3] += bvf[ni].w; }
                        uint2 pk;
                        pk.x = pack2bf(v[0] * gg[2 * ni].x, v[1] * gg[2 * ni].y);
                        pk.y = pack2bf(v[2] * gg[2 * ni + 1].x, v[3] * gg[2 * ni + 1].y);
                        *(uint2*)(wrow + ni * 32) = pk;
                    }
#else
#pragma unroll
                    for (int ni = 0; ni < 5; ++ni) {
                        f32x4 v = acc[mi][ni], g = acc[mi][ni + 5];
                        if (LNF) {          // t carries the bias
                            ln_fold4(v, ni, mi);
                            ln_fold4(g, ni + 5, mi);
                        } else if (p.bias) {
                            v[0] += bvf[ni].x; v[1] += bvf[ni].y; v[2] += bvf[ni].z; v[3] += bvf[ni].w;
                            g[0] += bvf[ni + 5].x; g[1] += bvf[ni + 5].y; g[2] += bvf[ni + 5].z; g[3] += bvf[ni + 5].w;
                        }
                        const wiw_f32x2 h01 = {v[0] * gelu_erf_f(g[0]), v[1] * gelu_erf_f(g[1])};
                        const wiw_f32x2 h23 = {v[2] * gelu_erf_f(g[2]), v[3] * gelu_erf_f(g[3])};
                        uint2 pk;
                        pk.x = pack2bf(h01.x, h01.y);
                        pk.y = pack2bf(h23.x, h23.y);
                        *(uint2*)(wrow + ni * 32) = pk;
                    }
#endif
                }
                wave_lds_sync();
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int slot = k * 64 + lane;           // 0..319
                    const int rr = slot / 10, ch = slot - rr * 10;
                    const int m = mw0 + rr;
                    const int nc = tile_n * 80 + ch * 8;
                    const bool ok = m < p.M && nc < n_valid;
                    const uint4 ov = *(const uint4*)(stg + rr * STG_ROWB_G + ch * 16);
                    uint4* dst = ok ? (uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + nc) : dump;
                    __builtin_nontemporal_store(ov.x, &dst->x); __builtin_nontemporal_store(ov.y, &dst->y);
                    __builtin_nontemporal_store(ov.z, &dst->z); __builtin_nontemporal_store(ov.w, &dst->w);
                }
                wave_lds_sync();
                if (fast_st) pending_stores = 5;
            } else {
                half_pass(std::integral_constant<bool, GE>{}, std::integral_constant<int, 0>{});
                half_pass(std::integral_constant<bool, GE>{}, std::integral_constant<int, 1>{});
                if (fast_st) pending_stores = 2 * ITEMS_P;
            }
#if WIW_ABLATE == 6
            } else {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 10; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
            }
#endif
        } else {
            // direct path (fp32 output, SiLU, unaligned N / strides): element-wise from the fragment layout.
            // Every index into acc[][] is a compile-time constant (guards instead of break / continue).
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = m0 + wave * 32 + mi * 16 + frow;
                const bool m_ok = m < p.M;
                const int mc = m_ok ? m : p.M - 1;
                const float* rv = p.rowvec ? p.rowvec + (int64_t)(mc / p.rows_per_vec) * p.rowvec_ld : nullptr;
#pragma unroll
                for (int ni = 0; ni < 10; ++ni) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = ni * 16 + fq * 4 + r;   // column inside the tile
                        float y = acc[mi][ni][r];
                        int n = n0 + c;
                        bool ok = m_ok;
                        if (geglu) {
                            n = tile_n * 80 + c;
                            ok = ok && ni < 5 && n < n_valid;
                            if (ok) {
                                const float bv = p.bias ? p.bias[n0 + c] : 0.f, bg = p.bias ? p.bias[n0 + 80 + c] : 0.f;
                                y = (y + bv) * gelu_erf_f(acc[mi][ni < 5 ? ni + 5 : ni][r] + bg);
                            }
                        } else {
                            ok = ok && n < n_valid;
                            if (ok) {
                                if (p.bias) y += p.bias[n];
                                if (rv) y += rv[n];
                                y *= p.alpha;
                                if (do_silu) y = silu_f(y);
                                if (act == WIW_EPI_GELU) y = gelu_erf_f(y);
                                else if (act == WIW_EPI_QUICK_GELU)   // x * sigmoid(1.702 x)
                                    y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * y));
                                if (r1) y += p.beta1 * (r1_f32 ? ((const float*)p.res1)[(int64_t)m * p.ldr1 + n] : bf2f(r1[(int64_t)m * p.ldr1 + n]));
                                if (r2) y += p.beta2 * (r2_f32 ? ((const float*)p.res2)[(int64_t)m * p.ldr2 + n] : bf2f(r2[(int64_t)m * p.ldr2 + n]));
                            }
                        }
                        if (ok) {
                            if (out_f32) ((float*)p.out)[out_slab + (int64_t)m * p.ldo + n] = y;
                            else ((uint16_t*)p.out)[(int64_t)m * p.ldo + n] = f2bf(y);
                        }
                    }
                }
            }
        }
        t = t_next;
        q = q_next;
#ifdef WIW_TRACE
        WIW_TP(2 + 8 * nk);      // end of the epilogue
        ++tiles_done;
#endif
    }
}

// split-K second pass: out = epilogue(sum over the S fp32 slabs, in slab order — deterministic), 8 columns per thread.
// Same arithmetic as the direct epilogue path: ((sum + bias + rowvec) * alpha [SiLU]) + beta1 res1 + beta2 res2.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const WiwGemmArgs p) {
    const int nch = p.N >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.M * nch) return;
    const int64_t m = i / nch;
    const int n = (int)(i - m * nch) * 8;
    const float* ws = (const float*)p.workspace + m * p.N + n;
    const int64_t slab = (int64_t)p.M * p.N;
    // round 6: EVERY load of the thread is issued before the first use (the slabs of up to four ranges, the bias / vector
    // rows as float4 pairs, the residual rows as one 16-byte or two 32-byte loads): the round 2-5 form walked the slabs in a
    // run-time loop (load, add, load, add ...) and fetched bias / residuals element by element — 32 us for the 93 MB of a
    // 4-range M = 4032, N = 1280 reduce (2.9 TB/s).  The sum order (slab 0, 1, 2 ...) and the epilogue arithmetic are unchanged.
    float4 sa[4], sb[4];
    const int S = p.splitk;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < S) { sa[k] = *(const float4*)(ws + k * slab); sb[k] = *(const float4*)(ws + k * slab + 4); }
    const bool r1_f32 = (p.epilogue & WIW_EPI_RES1_F32) != 0, r2_f32 = (p.epilogue & WIW_EPI_RES2_F32) != 0;
    float4 ba = float4{0.f, 0.f, 0.f, 0.f}, bb = ba, va = ba, vb = ba;
    if (p.bias) { ba = *(const float4*)(p.bias + n); bb = *(const float4*)(p.bias + n + 4); }
    if (p.rowvec) {
        const float* rv = p.rowvec + (m / p.rows_per_vec) * p.rowvec_ld + n;
        va = *(const float4*)rv; vb = *(const float4*)(rv + 4);
    }
    float f1[8], f2[8];
    if (p.res1) {
        if (r1_f32) {
            const float* r = (const float*)p.res1 + (m * p.ldr1 + n);
            const float4 a = *(const float4*)r, b = *(const float4*)(r + 4);
            f1[0] = a.x; f1[1] = a.y; f1[2] = a.z; f1[3] = a.w; f1[4] = b.x; f1[5] = b.y; f1[6] = b.z; f1[7] = b.w;
        } else {
            unpack8(*(const uint4*)((const uint16_t*)p.res1 + (m * p.ldr1 + n)), f1);
        }
    }
    if (p.res2) {
        if (r2_f32) {
            const float* r = (const float*)p.res2 + (m * p.ldr2 + n);
            const float4 a = *(const float4*)r, b = *(const float4*)(r + 4);
            f2[0] = a.x; f2[1] = a.y; f2[2] = a.z; f2[3] = a.w; f2[4] = b.x; f2[5] = b.y; f2[6] = b.z; f2[7] = b.w;
        } else {
            unpack8(*(const uint4*)((const uint16_t*)p.res2 + (m * p.ldr2 + n)), f2);
        }
    }
    float v[8] = {sa[0].x, sa[0].y, sa[0].z, sa[0].w, sb[0].x, sb[0].y, sb[0].z, sb[0].w};
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < S) {
            v[0] += sa[k].x; v[1] += sa[k].y; v[2] += sa[k].z; v[3] += sa[k].w;
            v[4] += sb[k].x; v[5] += sb[k].y; v[6] += sb[k].z; v[7] += sb[k].w;
        }
    for (int k = 4; k < S; ++k) {      // (more than four ranges: not used by the served network)
        const float4 a = *(const float4*)(ws + k * slab), b = *(const float4*)(ws + k * slab + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    const float bias8[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
    const float vec8[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
    const bool do_silu = (p.epilogue & WIW_EPI_SILU) != 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float y = v[e];
        if (p.bias) y += bias8[e];
        if (p.rowvec) y += vec8[e];
        y *= p.alpha;
        if (do_silu) y = silu_f(y);
        if (p.res1) y += p.beta1 * f1[e];
        if (p.res2) y += p.beta2 * f2[e];
        v[e] = y;
    }
    if (p.epilogue & WIW_EPI_OUT_F32) {
        float* o = (float*)p.out + m * p.ldo + n;
        *(float4*)o = float4{v[0], v[1], v[2], v[3]};
        *(float4*)(o + 4) = float4{v[4], v[5], v[6], v[7]};
    } else {
        *(uint4*)((uint16_t*)p.out + m * p.ldo + n) = pack8(v);
    }
}

template <int MODE, int NW, int STAGES, bool GE, bool LNF = false, bool F32E = false>
int launch_cfg(hipStream_t s, const WiwGemmArgs& a, int blocks_per_cu) {
    constexpr int BM = NW * 32;
    constexpr int SMEM = STAGES * (BM * BK * 2 + B_BYTES) + (LNF ? 2 * 2 * BN * 4 : 0);   // + two parities of s | t
    // one-time, thread-safe setup per template instantiation: opt in to > 64 KiB of dynamic LDS, read the CU count
    static std::once_flag once;
    static bool attr_ok = false;
    static int num_cu = 256;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute((const void*)gemm_kernel<MODE, NW, STAGES, GE, LNF, F32E>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
    });
    if (!attr_ok) {
        wiw_set_error("hipFuncSetAttribute(gemm) failed");
        return WIW_ELAUNCH;
    }
    const int64_t tiles = (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * (a.splitk > 1 ? a.splitk : 1);
    int64_t grid = (int64_t)num_cu * blocks_per_cu;   // persistent: every CU slot gets one block
    if (tiles < grid) grid = tiles >= 64 ? (tiles / 8) * 8 : tiles;   // keep the per-XCD super-tile schedule usable
    static const char* stg_env = getenv("WIW_GEMM_STAGGER");
    const int stagger = stg_env ? atoi(stg_env) : 0;
    hipLaunchKernelGGL((gemm_kernel<MODE, NW, STAGES, GE, LNF, F32E>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, a, stagger);
    return wiw_check_launch("wiw_gemm_bf16");
}

// BIG (256-row, 3-stage, 1 block/CU) unless M is too small to fill the chip or would waste > 10 % more rows
inline bool use_big_tile(const WiwGemmArgs& a) {
    const int64_t Nt = (a.N + BN - 1) / BN;
    const int64_t mt256 = (a.M + 255) / 256, mt128 = (a.M + 127) / 128;
    // (exactly one 256-row tile per CU — M = 8064, N = 1280 of the fine-tuning step: 32 x 8 = 256 tiles — is the best case of
    // the 256-row tile, 37 us against 60 us for the 128-row one; below that the finer tile fills more CUs)
    if (mt256 * Nt < 256) return false;
    return (double)(mt256 * 256) <= 1.1 * (double)(mt128 * 128);
}

template <int MODE>
int launch(hipStream_t s, const WiwGemmArgs& a) {
    if (a.splitk > 1) {   // pass 1: raw fp32 partial sums of every K range on the 256x160 tile; pass 2: reduce + epilogue
        WiwGemmArgs g = a;
        g.out = a.workspace; g.ldo = a.N;
        g.bias = nullptr; g.rowvec = nullptr; g.res1 = nullptr; g.res2 = nullptr;
        g.alpha = 1.0f; g.epilogue = WIW_EPI_OUT_F32 | (a.epilogue & (WIW_W_TILED | WIW_K_CMAJOR));   // (the fp32-residual bits belong to pass 2)
        // tile: 256 x 320 when N fills it, every K range keeps >= 10 K tiles and the ranges give (nearly) every CU an
        // item; else 256 x 160.   WIW_GEMM_TILE=huge|big overrides (A/B).
        static const char* force_sk = getenv("WIW_GEMM_TILE");
        const int64_t tiles_h = (int64_t)((a.M + 255) / 256) * ((a.N + 319) / 320) * a.splitk;
        bool huge = a.N % 320 == 0 && a.K / a.splitk >= 640 && tiles_h >= 200;
        if (force_sk) huge = force_sk[0] == 'h' && a.N % 320 == 0;
        const int rc = huge ? wiw_gemm_huge_launch(s, g) : launch_cfg<MODE, 8, 3, false>(s, g, 1);
        if (rc != WIW_OK) return rc;
        const int64_t n_thr = (int64_t)a.M * (a.N >> 3);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, s, a);
        return wiw_check_launch("wiw_gemm_bf16 (split-K reduce)");
    }
    static const char* force = getenv("WIW_GEMM_TILE");   // tuning knob: "huge" / "big" / "small" overrides the heuristic
    if (a.epilogue & WIW_EPI_LNFOLD) {   // whole rows of x per wave: the 256x160 / 128x160 tiles only (dense mode)
        if constexpr (MODE == WIW_A_DENSE) {
            const bool big_l = force ? (force[0] != 's') : use_big_tile(a);
            if (a.epilogue & WIW_EPI_GEGLU)
                return big_l ? launch_cfg<WIW_A_DENSE, 8, 3, true, true>(s, a, 1) : launch_cfg<WIW_A_DENSE, 4, 2, true, true>(s, a, 2);
            return big_l ? launch_cfg<WIW_A_DENSE, 8, 3, false, true>(s, a, 1) : launch_cfg<WIW_A_DENSE, 4, 2, false, true>(s, a, 2);
        } else {
            wiw_set_error("gemm: the LayerNorm fold is a dense-mode epilogue");
            return WIW_EINVAL;
        }
    }
    // fp32 residual stream (ABI 11): the 256x320 tile's F32E instantiations where that tile applies (round 4, second half),
    // else the F32E instantiation of this file on aligned shapes (everything else with these bits: the scalar direct path of
    // the common instantiation)
    if ((a.epilogue & (WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32 | WIW_EPI_OUT_F32)) && (!force || force[0] == 'h') && wiw_gemm_huge_ok(a))
        return wiw_gemm_huge_launch(s, a);
    if ((a.epilogue & (WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32 | WIW_EPI_OUT_F32)) &&
        !(a.epilogue & (WIW_EPI_GEGLU | WIW_EPI_SILU | WIW_EPI_GELU | WIW_EPI_QUICK_GELU)) && a.N % 4 == 0 && a.ldo % 4 == 0 &&
        (a.res1 == nullptr || a.ldr1 % 4 == 0) && (a.res2 == nullptr || a.ldr2 % 4 == 0) &&
        ((((uintptr_t)a.bias | (uintptr_t)a.rowvec) & 15) == 0) && a.rowvec_ld % 4 == 0) {
        const bool big_f = force ? (force[0] != 's') : use_big_tile(a);
        return big_f ? launch_cfg<MODE, 8, 3, false, false, true>(s, a, 1) : launch_cfg<MODE, 4, 2, false, false, true>(s, a, 2);
    }
    if ((!force || force[0] == 'h') && wiw_gemm_huge_ok(a)) return wiw_gemm_huge_launch(s, a);   // gemm_huge.hip
    const bool big = force ? (force[0] != 's') : use_big_tile(a);
    if (MODE == WIW_A_DENSE && (a.epilogue & WIW_EPI_GEGLU))
        return big ? launch_cfg<WIW_A_DENSE, 8, 3, true>(s, a, 1) : launch_cfg<WIW_A_DENSE, 4, 2, true>(s, a, 2);
    return big ? launch_cfg<MODE, 8, 3, false>(s, a, 1) : launch_cfg<MODE, 4, 2, false>(s, a, 2);
}

}  // namespace

#ifdef WIW_TRACE
extern "C" int wiw_gemm_trace_read(long long* out) {   // debug builds only: copy the slot-boundary timestamps to the host
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(long long) * 2 * 4096) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int wiw_gemm_bf16(void* stream, const WiwGemmArgs* args) {
    WIW_REQUIRE(args != nullptr, "gemm: null args");
    const WiwGemmArgs& a = *args;
    WIW_REQUIRE(a.A && a.W && a.out && a.zeros, "gemm: null A/W/out/zeros pointer");
    WIW_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: M, N, K must be positive");
    WIW_REQUIRE(a.C1 > 0 && a.C1 % 64 == 0 && a.C2 >= 0 && a.C2 % 64 == 0 && a.C3 >= 0 && a.C3 % 64 == 0,
                "gemm: C1/C2/C3 must be multiples of 64");
    WIW_REQUIRE((a.C2 == 0) == (a.A2 == nullptr), "gemm: A2 must be given iff C2 > 0");
    WIW_REQUIRE((a.C3 == 0) == (a.A3 == nullptr), "gemm: A3 must be given iff C3 > 0");
    WIW_REQUIRE(a.C3 == 0 || (a.mode == WIW_A_CONV3X3 && a.C2 > 0), "gemm: A3 only as the second half of a conv3x3 shortcut segment");
    WIW_REQUIRE(a.mode >= WIW_A_DENSE && a.mode <= WIW_A_CONV3X3_S2P, "gemm: unknown mode");
    const int taps = a.mode == WIW_A_DENSE ? 1 : (a.mode == WIW_A_CONV_T3 ? 3 : 9);
    WIW_REQUIRE(a.mode == WIW_A_DENSE || a.mode == WIW_A_CONV3X3 || a.C2 == 0, "gemm: A2 only in dense / conv3x3 mode");
    if (a.mode == WIW_A_DENSE) WIW_REQUIRE(a.K == a.C1 + a.C2, "gemm: K != C1 + C2");
    else WIW_REQUIRE(a.K == taps * a.C1 + a.C2 + a.C3, "gemm: K != taps * C1 (+ C2 + C3 shortcut segment)");
    if (a.mode != WIW_A_DENSE) {
        WIW_REQUIRE(a.H > 0 && a.Wd > 0, "gemm: conv geometry missing");
        WIW_REQUIRE(a.M % (a.H * a.Wd) == 0, "gemm: M must be a multiple of H*W in conv modes");
        if (a.mode == WIW_A_CONV3X3_UP) WIW_REQUIRE(a.H % 2 == 0 && a.Wd % 2 == 0, "gemm: upsample needs even H, W");
        if (a.mode == WIW_A_CONV_T3) WIW_REQUIRE(a.T > 0 && (a.M / (a.H * a.Wd)) % a.T == 0, "gemm: bad T");
    }
    WIW_REQUIRE(a.rowvec == nullptr || a.rows_per_vec > 0, "gemm: rows_per_vec must be > 0 with rowvec");
    WIW_REQUIRE(a.ldo > 0, "gemm: ldo must be positive");
    if (a.epilogue & (WIW_EPI_RES1_F32 | WIW_EPI_RES2_F32)) {
        WIW_REQUIRE(!(a.epilogue & (WIW_EPI_GEGLU | WIW_EPI_LNFOLD)), "gemm: fp32 residuals do not go with GEGLU / LayerNorm-fold epilogues");
        WIW_REQUIRE(!(a.epilogue & WIW_EPI_RES1_F32) || a.res1 != nullptr, "gemm: WIW_EPI_RES1_F32 without res1");
        WIW_REQUIRE(!(a.epilogue & WIW_EPI_RES2_F32) || a.res2 != nullptr, "gemm: WIW_EPI_RES2_F32 without res2");
    }
    if (a.epilogue & WIW_EPI_GEGLU) {
        WIW_REQUIRE(a.N % BN == 0 && a.n_out > 0 && a.n_out <= a.N / 2, "gemm: GEGLU needs N % 160 == 0 and n_out");
        WIW_REQUIRE(a.rowvec == nullptr && a.res1 == nullptr && a.res2 == nullptr, "gemm: GEGLU takes bias only");
        WIW_REQUIRE((((uintptr_t)a.bias) & 15) == 0, "gemm: GEGLU bias must be 16-byte aligned");
        WIW_REQUIRE(a.mode == WIW_A_DENSE, "gemm: GEGLU only in dense mode");
    }
    if (a.epilogue & WIW_EPI_LNFOLD) {
        WIW_REQUIRE(a.mode == WIW_A_DENSE && a.C2 == 0 && a.K == a.C1, "gemm: LayerNorm fold needs dense mode over one source (K = C1)");
        WIW_REQUIRE(a.lnfold != nullptr && (((uintptr_t)a.lnfold) & 15) == 0 && a.N % 160 == 0,
                    "gemm: LayerNorm fold needs lnfold = [s | t] (16-byte aligned) and N % 160 == 0");
        WIW_REQUIRE(a.bias == nullptr && a.alpha == 1.0f, "gemm: LayerNorm fold carries the bias in t and takes alpha = 1");
        WIW_REQUIRE(!(a.epilogue & (WIW_EPI_OUT_F32 | WIW_EPI_SILU | WIW_EPI_GELU | WIW_EPI_QUICK_GELU)) && a.splitk <= 1,
                    "gemm: LayerNorm fold goes with the staged 16-bit epilogue only");
        const int nv = (a.epilogue & WIW_EPI_GEGLU) ? a.n_out : a.N;
        WIW_REQUIRE(nv % 8 == 0 && a.ldo % 8 == 0 && (a.res1 == nullptr || a.ldr1 % 8 == 0) && (a.res2 == nullptr || a.ldr2 % 8 == 0) &&
                    (((uintptr_t)a.rowvec) & 15) == 0 && a.rowvec_ld % 4 == 0,
                    "gemm: LayerNorm fold needs the aligned (staged) output layout");
        WIW_REQUIRE(a.ln_eps > 0.0f, "gemm: LayerNorm fold needs ln_eps > 0");
    }
    if (a.epilogue & WIW_K_CMAJOR)
        WIW_REQUIRE(a.mode != WIW_A_DENSE, "gemm: WIW_K_CMAJOR is a convolution-mode K order");
    if (a.epilogue & WIW_K_HALO32)      // W's K order belongs to one kernel: checked and launched there, no other tile takes it
        return wiw_gemm_huge_launch((hipStream_t)stream, a);
    if (a.splitk > 1) {
        WIW_REQUIRE(a.workspace != nullptr, "gemm: split-K needs a workspace of splitk * M * N floats");
        WIW_REQUIRE((a.K / 64) % a.splitk == 0, "gemm: split-K needs K / 64 divisible by splitk");
        WIW_REQUIRE(!(a.epilogue & (WIW_EPI_GEGLU | WIW_EPI_GELU | WIW_EPI_QUICK_GELU)), "gemm: split-K does not take GEGLU / GELU epilogues");
        WIW_REQUIRE(a.N % 8 == 0 && a.ldo % 8 == 0 && (a.res1 == nullptr || a.ldr1 % 8 == 0) && (a.res2 == nullptr || a.ldr2 % 8 == 0),
                    "gemm: split-K needs N, ldo, ldr1, ldr2 multiples of 8");
        WIW_REQUIRE((((uintptr_t)a.bias | (uintptr_t)a.rowvec | (uintptr_t)a.res1 | (uintptr_t)a.res2 | (uintptr_t)a.out) & 15) == 0 &&
                    (a.rowvec == nullptr || a.rowvec_ld % 4 == 0),
                    "gemm: split-K needs 16-byte aligned bias / rowvec / residual / out pointers and rowvec_ld a multiple of 4");
    }
    hipStream_t s = (hipStream_t)stream;
    switch (a.mode) {
        case WIW_A_DENSE: return launch<WIW_A_DENSE>(s, a);
        case WIW_A_CONV3X3: return launch<WIW_A_CONV3X3>(s, a);
        case WIW_A_CONV3X3_S2: return launch<WIW_A_CONV3X3_S2>(s, a);
        case WIW_A_CONV3X3_UP: return launch<WIW_A_CONV3X3_UP>(s, a);
        case WIW_A_CONV3X3_S2P: return launch<WIW_A_CONV3X3_S2P>(s, a);
        default: return launch<WIW_A_CONV_T3>(s, a);
    }
}
