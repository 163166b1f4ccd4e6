// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   acc[m][n] = sum_k Agather[m][k] * W[n][k]      (both operands K-contiguous: "A . B^T")
//
// Design:
//   * two tile configurations of one kernel template:
//       BIG   : 256 (M) x 160 (N) x 64 (K), 512 threads = 8 waves as 4x2, 3-stage LDS ring (156 KiB,
//               one block per CU), two K tiles of LDS-DMA in flight behind the MFMAs (counted vmcnt);
//       SMALL : 128 x 160 x 64, 256 threads = 4 waves as 2x2, 2 stages (72 KiB, two blocks per CU) —
//               used when M is small (L3 / embeddings) so the grid still covers the 256 CUs.
//     Each wave owns 64x80 = 4x5 fragments of v_mfma_f32_16x16x32_bf16 (80 fp32 accumulator
//     VGPRs/lane).  BN = 160 because every channel count of the served UNet is a multiple of 320.
//   * operands go HBM -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip); ONE raw
//     s_barrier per K tile; waits are counted (`s_waitcnt vmcnt(N)`, never a drain while a later tile
//     is still needed in flight).  The implicit-GEMM gather (3x3 / stride-2 / nearest-upsample /
//     temporal taps, zero padding) is done on the per-lane SOURCE address — padding taps read a
//     16-byte zero buffer.
//   * LDS image: row r = 128 B (64 bf16); the 16-byte chunk c of row r is stored at chunk position
//     c ^ (r & 7).  With LDS-DMA the destination is lane-linear, so the swizzle is applied to the
//     source chunk each lane fetches and again on the ds_read_b128 — conflict-free fragment reads.
//   * epilogue: accumulators are staged through LDS so that bias / per-frame vector / residual reads
//     and the bf16 store are 16 B per lane on full rows; GEGLU pairs column c with column c+80 of the
//     same tile (weights are packed that way on the host).
//   * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs (private L2) gets a contiguous
//     range of tiles, N fastest, so an XCD's resident blocks share A and W panels in its L2.
#include "common.h"

namespace {

constexpr int BN = 160, BK = 64;
constexpr int B_BYTES = BN * BK * 2;              // 20480
constexpr int EP_LD = 164;                        // fp32 staging row stride

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

WIW_DEV void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int MODE, int WM_, int STAGES>
__global__ __launch_bounds__(WM_ * 128, 2) void gemm_kernel(const WiwGemmArgs p) {
    constexpr int BM = WM_ * 64;
    constexpr int NW = WM_ * 2;                       // waves per block
    constexpr int NT = NW * 64;                       // threads per block
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_FULL = 20 / NW;                   // full (8-row) W DMA instructions per wave
    constexpr bool B_HALF = (20 % NW) != 0;           // plus one 4-row instruction per wave (NW = 8)
    constexpr int LPT = 4 + B_FULL + (B_HALF ? 1 : 0);  // LDS-DMA instructions per wave per K tile
    static_assert(B_FULL * NW + (B_HALF ? NW / 2 : 0) == 20, "W tile must be covered exactly");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int Nt = (p.N + BN - 1) / BN;
    const int nb = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % Nt, tile_m = bid / Nt;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread source rows.  One wave instruction of global_load_lds moves 8 rows x 128 B.
    const int rsub = lane >> 3;                       // row inside the 8-row group
    const int chunk = (lane & 7) ^ (rsub & 7);        // logical 16-B chunk fetched into position lane&7
    const char* const Ab = (const char*)p.A;
    const char* const A2b = (const char*)p.A2;
    const char* const zeros = (const char*)p.zeros;
    const int HW = p.H * p.Wd;

    int a_m[4];          // global output row
    bool a_ok[4];
    int a_fb[4], a_y[4], a_x[4];   // conv: input-frame base row, output y / x  (mode 4: a_y = t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wave * 4 + i) * 8 + rsub;
        a_m[i] = m;
        a_ok[i] = m < p.M;
        a_fb[i] = 0; a_y[i] = 0; a_x[i] = 0;
        if (MODE == WIW_A_CONV3X3 || MODE == WIW_A_CONV3X3_S2 || MODE == WIW_A_CONV3X3_UP) {
            const int n = m / HW, rem = m - n * HW;
            a_y[i] = rem / p.Wd;
            a_x[i] = rem - a_y[i] * p.Wd;
            a_fb[i] = (MODE == WIW_A_CONV3X3) ? n * HW : (MODE == WIW_A_CONV3X3_S2 ? n * HW * 4 : n * (HW >> 2));
        } else if (MODE == WIW_A_CONV_T3) {
            a_y[i] = (m / HW) % p.T;
        }
    }
    const char* w_row[B_FULL + 1];
#pragma unroll
    for (int i = 0; i < B_FULL; ++i) {
        int n = n0 + (wave * B_FULL + i) * 8 + rsub;
        n = n < p.N ? n : p.N - 1;   // clamped rows are computed but never stored
        w_row[i] = (const char*)p.W + ((int64_t)n * p.K + chunk * 8) * 2;
    }
    {   // NW = 8: rows 128..159 of the W tile are fetched 4 rows per wave by lanes 0..31
        const int r_h = B_FULL * NW * 8 + wave * 4 + (rsub & 3);   // tile row; r_h & 7 != rsub for odd waves
        const int chunk_h = (lane & 7) ^ (r_h & 7);
        int n = n0 + r_h;
        n = n < p.N ? n : p.N - 1;
        w_row[B_FULL] = (const char*)p.W + ((int64_t)n * p.K + chunk_h * 8) * 2;
    }

    // address of the 16 bytes lane fetches for A row i, K tile (tap, cc)
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
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            int row;
            bool ok;
            if (MODE == WIW_A_CONV3X3) {
                const int iy = a_y[i] + dy, ix = a_x[i] + dx;
                ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                row = a_fb[i] + iy * p.Wd + ix;
            } else if (MODE == WIW_A_CONV3X3_S2) {
                const int iy = 2 * a_y[i] + dy, ix = 2 * a_x[i] + dx;
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

    const int Ctot = p.C1 + p.C2;
    auto issue = [&](int stage, int kt, int tap, int cc) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4 * 1024;
        char* sB = smem + stage * STAGE_BYTES + A_BYTES + wave * B_FULL * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src(i, tap, cc), sA + i * 1024);
#pragma unroll
        for (int i = 0; i < B_FULL; ++i) glds16(w_row[i] + (int64_t)kt * (BK * 2), sB + i * 1024);
        if (B_HALF) {
            char* sH = smem + stage * STAGE_BYTES + A_BYTES + B_FULL * NW * 1024 + wave * 512;
            if (lane < 32) glds16(w_row[B_FULL] + (int64_t)kt * (BK * 2), sH);
        }
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 5; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const char* sA = smem + stage * STAGE_BYTES + (wm * 64 + frow) * 128;
        const char* sB = smem + stage * STAGE_BYTES + A_BYTES + (wn * 80 + frow) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int sw = ((kk * 4 + fq) ^ (lane & 7)) << 4;
            bf16x8 a[4], b[5];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *(const bf16x8*)(sA + mi * 2048 + sw);
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) b[ni] = *(const bf16x8*)(sB + ni * 2048 + sw);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 5; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    };

    // ---- main loop: STAGES-1 K tiles of LDS-DMA in flight ahead of the MFMAs; one barrier per K tile.
    //   iteration kt:  wait(tile kt landed)  ->  barrier  ->  issue(tile kt+D into the stage that
    //   compute(kt-1) just released)  ->  compute(tile kt).  The barrier orders (a) every wave's DMA
    //   portion of tile kt before anybody's ds_read of it and (b) everybody's reads of stage (kt-1)
    //   before it is overwritten.
    constexpr int D = STAGES - 1;
    const int nk = p.K / BK;
    int tap = 0, cc = 0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        if (j < nk) {
            issue(j, j, tap, cc);
            cc += BK;
            if (cc == Ctot) { cc = 0; ++tap; }
        }
    }
    int st_c = 0, st_i = D % STAGES;   // stage computed / stage issued into
    for (int kt = 0; kt < nk; ++kt) {
        if (D == 2 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + D < nk) {
            issue(st_i, kt + D, tap, cc);
            cc += BK;
            if (cc == Ctot) { cc = 0; ++tap; }
        }
        compute(st_c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads are done before it arrives at the next barrier
        st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
        st_i = (st_i + 1 == STAGES) ? 0 : st_i + 1;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- epilogue through LDS, 64 rows per pass
    float* st = (float*)smem;
    const bool geglu = (p.epilogue & WIW_EPI_GEGLU) != 0;
    const bool do_silu = (p.epilogue & WIW_EPI_SILU) != 0;
    const bool out_f32 = (p.epilogue & WIW_EPI_OUT_F32) != 0;
    const int n_valid = geglu ? p.n_out : p.N;
    const bool vec_ok = (n_valid % 8 == 0) && (p.ldo % 8 == 0) && (p.res1 == nullptr || p.ldr1 % 8 == 0) &&
                        (p.res2 == nullptr || p.ldr2 % 8 == 0);
    const bool vec_rv = (((uintptr_t)p.bias | (uintptr_t)p.rowvec) & 15) == 0 && (p.rowvec_ld % 4 == 0);
    const int CH = geglu ? 10 : 20;
    const uint16_t* r1 = (const uint16_t*)p.res1;
    const uint16_t* r2 = (const uint16_t*)p.res2;
    // rows staged per pass: as many 64-row wave bands as fit in the LDS ring
    constexpr int ROWS_PP = (STAGES * STAGE_BYTES >= 128 * EP_LD * 4) ? 128 : 64;
    constexpr int PASSES = BM / ROWS_PP;
#pragma unroll 1
    for (int half = 0; half < PASSES; ++half) {
        if (half) __syncthreads();
        if ((wm * 64) / ROWS_PP == half) {
            const int rbase = (wm * 64) % ROWS_PP;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 5; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        st[(rbase + mi * 16 + fq * 4 + r) * EP_LD + wn * 80 + ni * 16 + frow] = acc[mi][ni][r];
        }
        __syncthreads();
        for (int item = tid; item < ROWS_PP * CH; item += NT) {
            const int row = item / CH, ch = item - row * CH;
            const int m = m0 + half * ROWS_PP + row;
            if (m >= p.M) continue;
            const int cl = ch * 8;                             // column inside the tile
            const int ncol = geglu ? tile_n * 80 + cl : n0 + cl;  // output column
            if (ncol >= n_valid) continue;
            float v[8];
            {
                const float4 t0 = *(const float4*)(st + row * EP_LD + cl);
                const float4 t1 = *(const float4*)(st + row * EP_LD + cl + 4);
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
                v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            }
            if (geglu) {
                float g[8];
                const float4 t0 = *(const float4*)(st + row * EP_LD + 80 + cl);
                const float4 t1 = *(const float4*)(st + row * EP_LD + 80 + cl + 4);
                g[0] = t0.x; g[1] = t0.y; g[2] = t0.z; g[3] = t0.w;
                g[4] = t1.x; g[5] = t1.y; g[6] = t1.z; g[7] = t1.w;
                if (p.bias) {   // packed bias: 16-byte aligned runs of 8 (n0, cl multiples of 8)
                    const float4 bv0 = *(const float4*)(p.bias + n0 + cl), bv1 = *(const float4*)(p.bias + n0 + cl + 4);
                    const float4 bg0 = *(const float4*)(p.bias + n0 + 80 + cl), bg1 = *(const float4*)(p.bias + n0 + 80 + cl + 4);
                    v[0] += bv0.x; v[1] += bv0.y; v[2] += bv0.z; v[3] += bv0.w;
                    v[4] += bv1.x; v[5] += bv1.y; v[6] += bv1.z; v[7] += bv1.w;
                    g[0] += bg0.x; g[1] += bg0.y; g[2] += bg0.z; g[3] += bg0.w;
                    g[4] += bg1.x; g[5] += bg1.y; g[6] += bg1.z; g[7] += bg1.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_f(g[e]);
            } else {
                const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.rowvec_ld : nullptr;
                if (vec_ok && vec_rv) {
                    if (p.bias) {
                        const float4 b0 = *(const float4*)(p.bias + ncol), b1 = *(const float4*)(p.bias + ncol + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                    if (rv) {
                        const float4 b0 = *(const float4*)(rv + ncol), b1 = *(const float4*)(rv + ncol + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int n = ncol + e;
                        if (n < n_valid) {
                            if (p.bias) v[e] += p.bias[n];
                            if (rv) v[e] += rv[n];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = v[e] * p.alpha;
                    if (do_silu) y = silu_f(y);
                    v[e] = y;
                }
            }
            if (vec_ok) {
                if (r1) {
                    float f[8];
                    unpack8(*(const uint4*)(r1 + (int64_t)m * p.ldr1 + ncol), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.beta1 * f[e];
                }
                if (r2) {
                    float f[8];
                    unpack8(*(const uint4*)(r2 + (int64_t)m * p.ldr2 + ncol), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.beta2 * f[e];
                }
                if (out_f32) {
                    float* o = (float*)p.out + (int64_t)m * p.ldo + ncol;
                    *(float4*)o = float4{v[0], v[1], v[2], v[3]};
                    *(float4*)(o + 4) = float4{v[4], v[5], v[6], v[7]};
                } else {
                    *(uint4*)((uint16_t*)p.out + (int64_t)m * p.ldo + ncol) = pack8(v);
                }
            } else {
                for (int e = 0; e < 8; ++e) {
                    const int n = ncol + e;
                    if (n >= n_valid) break;
                    float y = v[e];
                    if (r1) y += p.beta1 * bf2f(r1[(int64_t)m * p.ldr1 + n]);
                    if (r2) y += p.beta2 * bf2f(r2[(int64_t)m * p.ldr2 + n]);
                    if (out_f32) ((float*)p.out)[(int64_t)m * p.ldo + n] = y;
                    else ((uint16_t*)p.out)[(int64_t)m * p.ldo + n] = f2bf(y);
                }
            }
        }
    }
}

template <int MODE, int WM_, int STAGES>
int launch_cfg(hipStream_t s, const WiwGemmArgs& a) {
    constexpr int BM = WM_ * 64;
    constexpr int SMEM = STAGES * (BM * BK * 2 + B_BYTES);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<MODE, WM_, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                SMEM) != hipSuccess) {
            wiw_set_error("hipFuncSetAttribute(gemm) failed");
            return WIW_ELAUNCH;
        }
        attr_set = true;
    }
    const int Mt = (a.M + BM - 1) / BM, Nt = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_kernel<MODE, WM_, STAGES>), dim3(Mt * Nt), dim3(WM_ * 128), SMEM, s, a);
    return wiw_check_launch("wiw_gemm_bf16");
}

// BIG (256-row, 3-stage, 1 block/CU) unless M is too small to fill the chip or would waste > 10 % more rows
inline bool use_big_tile(const WiwGemmArgs& a) {
    const int64_t Nt = (a.N + BN - 1) / BN;
    const int64_t mt256 = (a.M + 255) / 256, mt128 = (a.M + 127) / 128;
    if (mt256 * Nt < 384) return false;
    return (double)(mt256 * 256) <= 1.1 * (double)(mt128 * 128);
}

template <int MODE>
int launch(hipStream_t s, const WiwGemmArgs& a) {
    return use_big_tile(a) ? launch_cfg<MODE, 4, 3>(s, a) : launch_cfg<MODE, 2, 2>(s, a);
}

}  // namespace

extern "C" int wiw_gemm_bf16(void* stream, const WiwGemmArgs* args) {
    WIW_REQUIRE(args != nullptr, "gemm: null args");
    const WiwGemmArgs& a = *args;
    WIW_REQUIRE(a.A && a.W && a.out && a.zeros, "gemm: null A/W/out/zeros pointer");
    WIW_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: M, N, K must be positive");
    WIW_REQUIRE(a.C1 > 0 && a.C1 % 64 == 0 && a.C2 >= 0 && a.C2 % 64 == 0, "gemm: C1/C2 must be multiples of 64");
    WIW_REQUIRE((a.C2 == 0) == (a.A2 == nullptr), "gemm: A2 must be given iff C2 > 0");
    WIW_REQUIRE(a.mode >= WIW_A_DENSE && a.mode <= WIW_A_CONV_T3, "gemm: unknown mode");
    const int taps = a.mode == WIW_A_DENSE ? 1 : (a.mode == WIW_A_CONV_T3 ? 3 : 9);
    WIW_REQUIRE(a.mode == WIW_A_DENSE || a.C2 == 0, "gemm: concat input only in dense mode");
    WIW_REQUIRE(a.K == taps * (a.C1 + a.C2), "gemm: K != taps * (C1 + C2)");
    if (a.mode != WIW_A_DENSE) {
        WIW_REQUIRE(a.H > 0 && a.Wd > 0, "gemm: conv geometry missing");
        WIW_REQUIRE(a.M % (a.H * a.Wd) == 0, "gemm: M must be a multiple of H*W in conv modes");
        if (a.mode == WIW_A_CONV3X3_UP) WIW_REQUIRE(a.H % 2 == 0 && a.Wd % 2 == 0, "gemm: upsample needs even H, W");
        if (a.mode == WIW_A_CONV_T3) WIW_REQUIRE(a.T > 0 && (a.M / (a.H * a.Wd)) % a.T == 0, "gemm: bad T");
    }
    WIW_REQUIRE(a.rowvec == nullptr || a.rows_per_vec > 0, "gemm: rows_per_vec must be > 0 with rowvec");
    WIW_REQUIRE(a.ldo > 0, "gemm: ldo must be positive");
    if (a.epilogue & WIW_EPI_GEGLU) {
        WIW_REQUIRE(a.N % BN == 0 && a.n_out > 0 && a.n_out <= a.N / 2, "gemm: GEGLU needs N % 160 == 0 and n_out");
        WIW_REQUIRE(a.rowvec == nullptr && a.res1 == nullptr && a.res2 == nullptr, "gemm: GEGLU takes bias only");
    }
    hipStream_t s = (hipStream_t)stream;
    switch (a.mode) {
        case WIW_A_DENSE: return launch<WIW_A_DENSE>(s, a);
        case WIW_A_CONV3X3: return launch<WIW_A_CONV3X3>(s, a);
        case WIW_A_CONV3X3_S2: return launch<WIW_A_CONV3X3_S2>(s, a);
        case WIW_A_CONV3X3_UP: return launch<WIW_A_CONV3X3_UP>(s, a);
        default: return launch<WIW_A_CONV_T3>(s, a);
    }
}
