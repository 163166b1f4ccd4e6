// GroupNorm(32) in NHWC (statistics / finalize / fused apply + SiLU) and LayerNorm with fused
// pre-add.  HBM-bound kernels: every global access is 16 B per lane on channel-contiguous rows,
// reductions are deterministic (fixed-order sums through LDS / a block-order second stage; DPP and permlane
// swaps inside a wave), statistics are fp32.
#include "common.h"

namespace {

constexpr int GROUPS = 32;

// ---------------------------------------------------------------------------------------------
// statistics: grid = (row_splits, units).  A block reduces rows [r0, r1) of one unit for ALL channels.
// Threads are laid out (channel chunk, row lane); a thread keeps per-channel partial sums of its
// 8 channels in registers, so the inner loop has no group arithmetic.
// NUMERICALLY STABLE: a thread accumulates SHIFTED sums  sum(x - p), sum((x - p)^2)  around a per-channel pivot p (the
// first row it sees), turns them into (mean, M2 = sum (x - mean)^2) and every later stage MERGES (count, mean, M2)
// triples with Chan's pairwise formula — row lanes in lane order, channels of a group in channel order, blocks in block
// order.  The single-pass E[x^2] - mean^2 of round 1 lost the variance for |mean| >> std (torch uses Welford; trained
// SVD activations have outlier channels).
// DETERMINISTIC: no atomics anywhere — every merge order is fixed; every block writes its 32 (mean, M2) pairs to
// `partials[unit][block][32][2]`, and the second stage (below) merges the blocks in block order.  (With fp32 atomics the same
// request differed by 1.6e-2 relative rms between two runs after two Euler steps at full size: a random-init network
// amplifies last-bit differences; the reference's GroupNorm is deterministic.)
// ---------------------------------------------------------------------------------------------
constexpr int GN_MAXC = 4096;
#ifndef WIW_GN_U
#define WIW_GN_U 4
#endif
#ifndef WIW_GN_WAVES
#define WIW_GN_WAVES 5                  // waves per SIMD the register allocation is held to (96 VGPRs, no scratch)
#endif
constexpr int GN_U = WIW_GN_U;          // loads in flight per thread of the statistics pass (A/B: tools/build_variant.py)

// 8 consecutive channels starting at element offset `off` of a 16-bit (F32IN = false: one 16-byte load, returned packed)
// or fp32 (two 16-byte loads) tensor.  The fp32 form serves the fp32 residual stream (ABI 11).
template <bool F32IN> struct Raw8 { uint4 a; };
template <> struct Raw8<true> { float4 a, b; };
template <bool F32IN>
WIW_DEV Raw8<F32IN> load_raw8(const void* base, int64_t off) {
    Raw8<F32IN> r;
    if constexpr (F32IN) {
        const float* p = (const float*)base + off;
        r.a = *(const float4*)p; r.b = *(const float4*)(p + 4);
    } else {
        r.a = *(const uint4*)((const uint16_t*)base + off);
    }
    return r;
}
template <bool F32IN>
WIW_DEV void raw8_to_f(const Raw8<F32IN>& r, float* f) {
    if constexpr (F32IN) {
        f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w; f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w;
    } else {
        unpack8(r.a, f);
    }
}

// (na, ma, Ma) <- merge with (nb, mb, Mb);  counts are exact small integers held in fp32
WIW_DEV void chan_merge(float& na, float& ma, float& Ma, float nb, float mb, float Mb) {
    if (nb <= 0.f) return;
    if (na <= 0.f) { na = nb; ma = mb; Ma = Mb; return; }
    const float n = na + nb, d = mb - ma, f = nb * __builtin_amdgcn_rcpf(n);   // 1-ulp reciprocal: deterministic, far below the bf16 output
    ma = __builtin_fmaf(d, f, ma);
    Ma = Ma + Mb + d * d * na * f;
    na = n;
}

// SECOND STAGE IN THE SAME LAUNCH (round 4, ABI 12; the stand-alone gn_reduce_kernel cost 8.7 us x 105 launches per
// forward): blocks are grouped in PARTS of GN_PART consecutive blocks.  Every block publishes its partial and bumps its
// part's counter; the block that finds the part complete merges the part's blocks (thread = (group, 4 consecutive blocks),
// then the 8 sub-results in order) into `parts[unit][part]`, then bumps the unit's counter the same way; the block that
// finds the unit complete merges the parts in part order and writes stats[unit].  WHICH block does a merge depends on
// timing, WHAT it computes does not: every merge order is fixed.
// Publication WITHOUT agent-scope fences: `__threadfence()` is buffer_wbl2 + buffer_inv of the XCD's whole L2, and 2 000
// blocks doing that serialise chip-wide (measured, profiles/r10a_norm_probe.txt: 0.12 us per block, the pass at 0.7 TB/s).
// Instead the exchanged words themselves are agent-scope accesses — stores written through (sc1), completed
// (s_waitcnt vmcnt(0)) before the counter's atomic is issued; loads that bypass the non-coherent levels (sc0 sc1), issued
// after the atomic returned — so no cached copy of them exists anywhere.  The streamed tensor never needs coherence.
// counters: [units][nparts + 1] unsigned, zero before the launch, left at zero.
constexpr int GN_PART = 32;

WIW_DEV void gn_publish2(float* p, float x, float y) {       // 8-byte agent-scope store
    __hip_atomic_store((unsigned long long*)p, ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// N independent 8-byte (16-byte) loads that bypass L1 / L2 non-coherent copies, all in flight together, waited for inside
// the statement (the compiler sees plain register outputs)
WIW_DEV void gn_fetch4x2(const float2* p0, const float2* p1, const float2* p2, const float2* p3, float2* v) {
    unsigned long long a, b, c, d;
    asm volatile(
        "global_load_dwordx2 %0, %4, off sc0 sc1\n\t"
        "global_load_dwordx2 %1, %5, off sc0 sc1\n\t"
        "global_load_dwordx2 %2, %6, off sc0 sc1\n\t"
        "global_load_dwordx2 %3, %7, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
    const unsigned long long r[4] = {a, b, c, d};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = float2{__uint_as_float((unsigned)(r[i] & 0xffffffffull)), __uint_as_float((unsigned)(r[i] >> 32))};
}
WIW_DEV void gn_fetch4x4(const float4* p0, const float4* p1, const float4* p2, const float4* p3, float4* v) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u a, b, c, d;
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
    const v4u r[4] = {a, b, c, d};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v[i] = float4{__uint_as_float(r[i].x), __uint_as_float(r[i].y), __uint_as_float(r[i].z), __uint_as_float(r[i].w)};
}

template <bool F32IN>
__global__ __launch_bounds__(256, WIW_GN_WAVES) void gn_stats_kernel(const void* __restrict__ X1, int C1,
                                                        const void* __restrict__ X2, int C2, int rows_per_unit,
                                                        int rows_per_block, float* partials, unsigned* counters,
                                                        float* __restrict__ stats) {
    __shared__ float red[256][17];          // per-thread (mean[8], M2[8]); 17: skewed banks
    extern __shared__ float chan_dyn[];     // per-channel block (mean, M2): [2][C]
    const int tid = threadIdx.x;
    const int C = C1 + C2;
    float* const chan0 = chan_dyn;
    float* const chan1 = chan_dyn + C;
    const int cg = C / GROUPS;
    const int chunks = C >> 3;
    const int cpb = chunks < 256 ? chunks : 256;   // chunk columns handled per pass
    const int rp = 256 / cpb;                       // row lanes
    const int ci = tid % cpb, rl = tid / cpb;
    const int unit = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows_per_unit) r1 = rows_per_unit;
    const int64_t base_row = (int64_t)unit * rows_per_unit;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {      // uniform trip count: the barriers below are reached by all
        const int chunk = cbase + ci;
        const bool active = rl < rp && chunk < chunks && r0 + rl < r1;
        const int c0 = chunk * 8;
        // packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: two channels per VALU instruction) keeps this pass memory-bound:
        // per 16-byte load 8 unpack + 4 sub + 4 add + 4 fma instructions
        wiw_f32x2 s2[4], q2[4], p2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s2[j] = wiw_f32x2{0.f, 0.f}; q2[j] = s2[j]; p2[j] = s2[j]; }
        int cnt = 0;
        auto pairs = [&](const Raw8<F32IN>& raw, wiw_f32x2* v) {
            if constexpr (F32IN) {
                v[0] = wiw_f32x2{raw.a.x, raw.a.y}; v[1] = wiw_f32x2{raw.a.z, raw.a.w};
                v[2] = wiw_f32x2{raw.b.x, raw.b.y}; v[3] = wiw_f32x2{raw.b.z, raw.b.w};
            } else {
                v[0] = unpack2(raw.a.x); v[1] = unpack2(raw.a.y); v[2] = unpack2(raw.a.z); v[3] = unpack2(raw.a.w);
            }
        };
        auto accum = [&](const Raw8<F32IN>& raw) {
            wiw_f32x2 v[4];
            pairs(raw, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wiw_f32x2 d = v[j] - p2[j];
                s2[j] += d;
                q2[j] = __builtin_elementwise_fma(d, d, q2[j]);
            }
        };
        if (active) {
            const void* src;
            int ld, coff;
            if (c0 < C1) { src = X1; ld = C1; coff = c0; } else { src = X2; ld = C2; coff = c0 - C1; }
            int r = r0 + rl;
            pairs(load_raw8<F32IN>(src, (base_row + r) * ld + coff), p2);   // pivot = first row (re-read below: L1 hit)
            for (; r + (GN_U - 1) * rp < r1; r += GN_U * rp) {   // GN_U independent (pairs of) 16-byte loads in flight per thread
                Raw8<F32IN> raw[GN_U];
#pragma unroll
                for (int u = 0; u < GN_U; ++u) raw[u] = load_raw8<F32IN>(src, (base_row + r + u * rp) * ld + coff);
#pragma unroll
                for (int u = 0; u < GN_U; ++u) accum(raw[u]);
                cnt += GN_U;
            }
            for (; r < r1; r += rp) {
                accum(load_raw8<F32IN>(src, (base_row + r) * ld + coff));
                ++cnt;
            }
        }
        {
            const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float sv = s2[j][h], ms = sv * inv;           // mean of the shifted values
                    red[tid][2 * j + h] = p2[j][h] + ms;
                    red[tid][8 + 2 * j + h] = fmaxf(q2[j][h] - sv * ms, 0.f);    // M2 around the thread's own mean
                }
            }
        }
        __syncthreads();
        // merge of the row lanes of a channel, fixed order, all 256 threads (thread = (chunk column ci, element pair)):
        // mean = sum n_j mean_j / N, M2 = sum M2_j + sum n_j (mean_j - mean)^2  — the parallel-variance formula in its
        // two-pass form (no per-merge division; stable: every term is a non-negative second moment about the merged mean)
        for (int t = tid; t < cpb * 8; t += 256) {
            const int cc = t >> 3, e = t & 7;
            if (cbase + cc < chunks) {
                const float inv_n = 1.0f / (float)(r1 - r0);
                float ms = 0.f;
                for (int j = 0; j < rp; ++j) {
                    const int left = r1 - r0 - j;
                    const float nj = left > 0 ? (float)((left + rp - 1) / rp) : 0.f;   // rows row-lane j walked
                    ms += nj * red[j * cpb + cc][e];
                }
                const float mean = ms * inv_n;
                float M2 = 0.f;
                for (int j = 0; j < rp; ++j) {
                    const int left = r1 - r0 - j;
                    const float nj = left > 0 ? (float)((left + rp - 1) / rp) : 0.f;
                    const float dm = red[j * cpb + cc][e] - mean;
                    M2 += red[j * cpb + cc][8 + e] + nj * dm * dm;
                }
                chan0[(cbase + cc) * 8 + e] = mean;
                chan1[(cbase + cc) * 8 + e] = M2;
            }
        }
        __syncthreads();
    }
    if (tid < GROUPS) {   // channels of a group in channel order; every channel carries (r1 - r0) rows
        const int g = tid;
        const float nr = (float)(r1 - r0);
        float ms = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) ms += chan0[c];
        const float mean = ms / (float)cg;
        float M2 = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) { const float dm = chan0[c] - mean; M2 += chan1[c] + nr * dm * dm; }
        gn_publish2(partials + ((int64_t)unit * gridDim.x + blockIdx.x) * (GROUPS * 2) + 2 * g, mean, M2);
    }
    // ---- second stage (see the header of this section)
    __shared__ int s_flag;
    __shared__ float red2[8][GROUPS][3];
    const int splits = (int)gridDim.x;
    const int nparts = (splits + GN_PART - 1) / GN_PART;
    const int w = (int)blockIdx.x / GN_PART;
    const int b0 = w * GN_PART, b1 = b0 + GN_PART < splits ? b0 + GN_PART : splits;
    unsigned* cnt = counters + (int64_t)unit * (nparts + 1);
    float* parts = partials + (int64_t)gridDim.y * splits * (GROUPS * 2);      // [units][nparts][32][4]: (n, mean, M2, -)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's published words have landed (write-through acknowledged)
    __syncthreads();
    if (tid == 0) {
        s_flag = atomicAdd(&cnt[1 + w], 1u) == (unsigned)(b1 - b0 - 1) ? 1 : 0;
        // every block of the part has passed: nobody touches the counter again in this launch
        if (s_flag) __hip_atomic_store(&cnt[1 + w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_flag) return;
    auto elems_of = [&](int b) {
        const int left = rows_per_unit - b * rows_per_block;
        return (float)(left < rows_per_block ? left : rows_per_block) * (float)cg;
    };
    {
        const int g = tid & 31, j = tid >> 5;
        const float2* p = (const float2*)(partials + (int64_t)unit * splits * (GROUPS * 2)) + g;
        const int bl = b1 - 1;                                   // out-of-part slots re-read the last block (ignored)
        const int bb = b0 + 4 * j;
        float2 v[4];
        gn_fetch4x2(p + (int64_t)(bb < bl ? bb : bl) * GROUPS, p + (int64_t)(bb + 1 < bl ? bb + 1 : bl) * GROUPS,
                    p + (int64_t)(bb + 2 < bl ? bb + 2 : bl) * GROUPS, p + (int64_t)(bb + 3 < bl ? bb + 3 : bl) * GROUPS, v);
        float n = 0.f, m = 0.f, M2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i;
            if (b < b1) chan_merge(n, m, M2, elems_of(b), v[i].x, v[i].y);
        }
        red2[j][g][0] = n; red2[j][g][1] = m; red2[j][g][2] = M2;
    }
    __syncthreads();
    float n = 0.f, m = 0.f, M2 = 0.f;
    if (tid < GROUPS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) chan_merge(n, m, M2, red2[j][tid][0], red2[j][tid][1], red2[j][tid][2]);
        if (nparts > 1) {
            float* pr = parts + (((int64_t)unit * nparts + w) * GROUPS + tid) * 4;
            gn_publish2(pr, n, m);
            gn_publish2(pr + 2, M2, 0.f);
        }
    }
    if (nparts > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            s_flag = atomicAdd(&cnt[0], 1u) == (unsigned)(nparts - 1) ? 1 : 0;
            if (s_flag) __hip_atomic_store(&cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_flag) return;
        if (tid < GROUPS) {
            const float4* pp = (const float4*)(parts + ((int64_t)unit * nparts * GROUPS + tid) * 4);
            n = 0.f; m = 0.f; M2 = 0.f;
            const int wl = nparts - 1;
            for (int w0 = 0; w0 < nparts; w0 += 4) {       // part order; 4 loads in flight
                float4 u[4];
                gn_fetch4x4(pp + (int64_t)(w0 < wl ? w0 : wl) * GROUPS, pp + (int64_t)(w0 + 1 < wl ? w0 + 1 : wl) * GROUPS,
                            pp + (int64_t)(w0 + 2 < wl ? w0 + 2 : wl) * GROUPS, pp + (int64_t)(w0 + 3 < wl ? w0 + 3 : wl) * GROUPS, u);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (w0 + i < nparts) chan_merge(n, m, M2, u[i].x, u[i].y, u[i].z);
            }
        }
    }
    if (tid < GROUPS) {
        stats[(int64_t)unit * (GROUPS * 2) + 2 * tid] = m;
        stats[(int64_t)unit * (GROUPS * 2) + 2 * tid + 1] = n > 0.f ? M2 / n : 0.f;
    }
}

// ab[unit][0][c] = rstd*gamma[c];  ab[unit][1][c] = beta[c] - mean*rstd*gamma[c]
__global__ void gn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, int units, int C, float inv_count, float eps,
                                   float* ab) {
    const int total = units * C;
    const int cg = C / GROUPS;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int u = idx / C, c = idx - u * C;
        const int g = c / cg;
        const float mean = stats[(u * GROUPS + g) * 2];
        const float var = stats[(u * GROUPS + g) * 2 + 1];
        const float rstd = rsqrtf(var + eps);
        const float a = rstd * gamma[c];
        ab[((int64_t)u * 2) * C + c] = a;
        ab[((int64_t)u * 2 + 1) * C + c] = beta[c] - mean * a;
    }
}

// apply: grid = (row_splits, units) like the statistics kernel; a thread keeps the scale/shift of its 8 channels in
// registers and walks rows with 32-bit index arithmetic only (no per-element division).
// FUSED: `ab` is the raw statistics buffer [units][32][2]; scale / shift are derived here (no finalize launch).
template <bool FUSED, bool F32IN = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ X1, int C1,
                                                        const void* __restrict__ X2, int C2, int rows_per_unit,
                                                        int rows_per_block, const float* __restrict__ ab, int silu,
                                                        uint16_t* out, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float inv_count, float eps,
                                                        uint16_t* raw16) {
    const int tid = threadIdx.x;
    const int C = C1 + C2;
    const int chunks = C >> 3;
    const int cpb = chunks < 256 ? chunks : 256;
    const int rp = 256 / cpb;
    const int ci = tid % cpb, rl = tid / cpb;
    if (rl >= rp) return;
    const int unit = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows_per_unit) r1 = rows_per_unit;
    const int64_t base_row = (int64_t)unit * rows_per_unit;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {
        const int chunk = cbase + ci;
        if (chunk >= chunks) break;
        const int c0 = chunk * 8;
        const void* src;
        int ld, coff;
        if (c0 < C1) { src = X1; ld = C1; coff = c0; } else { src = X2; ld = C2; coff = c0 - C1; }
        float4 a0, a1, b0, b1;
        if (FUSED) {
            const int cg = C / GROUPS;
            float av[8], bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e, g = c / cg;
                const float mean = ab[((int64_t)unit * GROUPS + g) * 2];
                const float var = ab[((int64_t)unit * GROUPS + g) * 2 + 1];
                av[e] = rsqrtf(var + eps) * gamma[c];
                bv[e] = beta[c] - mean * av[e];
            }
            a0 = float4{av[0], av[1], av[2], av[3]}; a1 = float4{av[4], av[5], av[6], av[7]};
            b0 = float4{bv[0], bv[1], bv[2], bv[3]}; b1 = float4{bv[4], bv[5], bv[6], bv[7]};
        } else {
            const float* a = ab + ((int64_t)unit * 2) * C + c0;
            const float* b = a + C;
            a0 = *(const float4*)a; a1 = *(const float4*)(a + 4);
            b0 = *(const float4*)b; b1 = *(const float4*)(b + 4);
        }
        int r = r0 + rl;
        for (; r + 3 * rp < r1; r += 4 * rp) {   // four independent loads in flight per thread
            Raw8<F32IN> raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = load_raw8<F32IN>(src, (base_row + r + u * rp) * ld + coff);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                raw8_to_f<F32IN>(raw[u], f);
                if (F32IN && raw16) *(uint4*)(raw16 + (base_row + r + u * rp) * C + c0) = pack8(f);
                f[0] = f[0] * a0.x + b0.x; f[1] = f[1] * a0.y + b0.y; f[2] = f[2] * a0.z + b0.z; f[3] = f[3] * a0.w + b0.w;
                f[4] = f[4] * a1.x + b1.x; f[5] = f[5] * a1.y + b1.y; f[6] = f[6] * a1.z + b1.z; f[7] = f[7] * a1.w + b1.w;
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
                }
                *(uint4*)(out + (base_row + r + u * rp) * C + c0) = pack8(f);
            }
        }
        for (; r < r1; r += rp) {
            float f[8];
            raw8_to_f<F32IN>(load_raw8<F32IN>(src, (base_row + r) * ld + coff), f);
            if (F32IN && raw16) *(uint4*)(raw16 + (base_row + r) * C + c0) = pack8(f);
            f[0] = f[0] * a0.x + b0.x; f[1] = f[1] * a0.y + b0.y; f[2] = f[2] * a0.z + b0.z; f[3] = f[3] * a0.w + b0.w;
            f[4] = f[4] * a1.x + b1.x; f[5] = f[5] * a1.y + b1.y; f[6] = f[6] * a1.z + b1.z; f[7] = f[7] * a1.w + b1.w;
            if (silu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
            }
            *(uint4*)(out + (base_row + r) * C + c0) = pack8(f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(32) (+SiLU) in ONE pass for the small levels (round 5): a (frame, group) unit of the two inner levels is 576 x 40
// or 144 x 40 values — a block keeps 160 consecutive channels of ONE frame (4 groups of 40, or 2 of 80 on a 2560-channel
// concat) in its registers: read once, exact two-pass mean / variance in fp32 (no cancellation: the data is resident),
// normalise, SiLU, store.  The two-kernel path reads those tensors twice and at these sizes both of its launches are
// latency-bound (41 MB: 17 + 14 us at 18 x 32, ~10 + 8 us at 9 x 16), and unlike the MFMA kernels the HBM-bound ones do not
// pay the board's power management: what they save arrives.  240 threads = 12 rows x 20 chunks of 8 channels per
// pass, PASSES = rows / 12 (48 or 12) loads per thread, all in flight before the first use.  Deterministic: fixed reduction
// order through LDS, no atomics.  16-bit tensors only (the fp32 stream keeps the two-kernel path).
// ---------------------------------------------------------------------------------------------
constexpr int GN1_ROWS = 12, GN1_CHUNKS = 20, GN1_THREADS = GN1_ROWS * GN1_CHUNKS, GN1_CB = GN1_CHUNKS * 8;
template <int PASSES>
__global__ __launch_bounds__(GN1_THREADS) void gn_onepass_kernel(const uint16_t* __restrict__ X1, int C1, const uint16_t* __restrict__ X2, int C2,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 int silu, uint16_t* __restrict__ out) {
    __shared__ float red[GN1_ROWS][GN1_CHUNKS];
    __shared__ float colsum[GN1_CHUNKS];
    const int tid = threadIdx.x, cv = tid % GN1_CHUNKS, rl = tid / GN1_CHUNKS;
    const int C = C1 + C2, cpgv = (C / GROUPS) >> 3;          // chunks per group: 5 (C = 1280) or 10 (2560)
    const int c0 = blockIdx.x * GN1_CB;                       // first channel of this block: inside ONE source (C1 % 160 == 0)
    const int64_t row0 = (int64_t)blockIdx.y * (PASSES * GN1_ROWS);
    const uint16_t* src;
    int ld, coff;
    if (c0 < C1) { src = X1; ld = C1; coff = c0; } else { src = X2; ld = C2; coff = c0 - C1; }
    uint4 raw[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; ++i) raw[i] = *(const uint4*)(src + (row0 + i * GN1_ROWS + rl) * ld + coff + cv * 8);
    const float4 g0 = *(const float4*)(gamma + c0 + cv * 8), g1 = *(const float4*)(gamma + c0 + cv * 8 + 4);
    const float4 b0 = *(const float4*)(beta + c0 + cv * 8), b1 = *(const float4*)(beta + c0 + cv * 8 + 4);
    const float inv_n = 1.0f / (float)(PASSES * GN1_ROWS * cpgv * 8);
    const int gfirst = (cv / cpgv) * cpgv;                   // first chunk of this thread's group inside the block
    // sum of `v` over the thread's group (all rows of the frame): rows through LDS in row order, then the group's chunks in order
    auto group_sum = [&](float v) -> float {
        red[rl][cv] = v;
        __syncthreads();
        if (rl == 0) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < GN1_ROWS; ++r) t += red[r][cv];
            colsum[cv] = t;
        }
        __syncthreads();
        float g = 0.f;
        for (int k = 0; k < cpgv; ++k) g += colsum[gfirst + k];
        __syncthreads();
        return g;
    };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        float f[8];
        unpack8(raw[i], f);
        s += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    }
    const float mean = group_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        float f[8];
        unpack8(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q = __builtin_fmaf(d, d, q); }
    }
    const float rstd = rsqrtf(group_sum(q) * inv_n + eps);
    const float a[8] = {rstd * g0.x, rstd * g0.y, rstd * g0.z, rstd * g0.w, rstd * g1.x, rstd * g1.y, rstd * g1.z, rstd * g1.w};
    const float b[8] = {b0.x - mean * a[0], b0.y - mean * a[1], b0.z - mean * a[2], b0.w - mean * a[3],
                        b1.x - mean * a[4], b1.y - mean * a[5], b1.z - mean * a[6], b1.w - mean * a[7]};
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
        float f[8];
        unpack8(raw[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f[e] = __builtin_fmaf(f[e], a[e], b[e]);
            if (silu) f[e] = silu_f(f[e]);
        }
        *(uint4*)(out + (row0 + i * GN1_ROWS + rl) * C + c0 + cv * 8) = pack8(f);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 2048 -> <= 4 chunks of 8 per lane),
// exact two-pass mean / variance with 64-lane butterfly reductions.
// ---------------------------------------------------------------------------------------------
#ifndef LN_R_MID
#define LN_R_MID 2     // rows in flight per wave at 512 < C <= 1024 (A/B knob)
#endif
#ifndef LN_R_WIDE
#define LN_R_WIDE 1    // ... at C > 1024
#endif
constexpr int LN_MAXCH = 4;
constexpr int LN_RUN = 8;      // consecutive rows per wave

// NCH = 16-byte chunk passes per row (C <= NCH * 512), R = rows whose loads are in flight together (8 at C <= 512):
// at C = 320 one row is only 640 B; with 1 / 4 / 8 rows in flight per wave the kernel streams 2.7 / 3.0 / 4.3 TB/s.
template <int NCH, int R, bool F32IN = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ X, int64_t rows, int C,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, const float* __restrict__ addvec, int addvec_ld,
                                                         int rows_per_vec, uint16_t* sum_out, uint16_t* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int chunks = C >> 3;
    const float inv_c = 1.0f / (float)C;
    float4 g0[NCH], g1[NCH], b0[NCH], b1[NCH], a0[NCH], a1[NCH];   // per-lane slices of gamma / beta / the add vector
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int chunk = lane + k * 64;
        if (chunk < chunks) {
            g0[k] = *(const float4*)(gamma + chunk * 8); g1[k] = *(const float4*)(gamma + chunk * 8 + 4);
            b0[k] = *(const float4*)(beta + chunk * 8); b1[k] = *(const float4*)(beta + chunk * 8 + 4);
        }
    }
    int vi_loaded = -1;
    // every wave walks LN_RUN consecutive rows: the addvec row index is divided once per run and then only compared
    for (int64_t run0 = ((int64_t)blockIdx.x * 4 + wave) * LN_RUN; run0 < rows; run0 += (int64_t)gridDim.x * 4 * LN_RUN) {
      int vi = addvec ? (int)(run0 / rows_per_vec) : 0;
      int64_t vnext = addvec ? (int64_t)(vi + 1) * rows_per_vec : rows;
      const int64_t run1 = run0 + LN_RUN < rows ? run0 + LN_RUN : rows;
      for (int64_t rowg = run0; rowg < run1; rowg += R) {
        float v[R][NCH][8];
        Raw8<F32IN> raw[R][NCH];
        // ---- all loads of the R rows first
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = rowg + r < run1 ? rowg + r : run1 - 1;   // tail rows re-read the last row (not stored)
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int chunk = lane + k * 64;
                if (chunk < chunks) raw[r][k] = load_raw8<F32IN>(X, row * C + chunk * 8);
            }
        }
        float s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = rowg + r;
            const bool live = row < run1;
            if (live && row >= vnext) { ++vi; vnext += rows_per_vec; }
            if (addvec && vi != vi_loaded) {   // wave-uniform; once per rows_per_vec rows
                const float* av = addvec + (int64_t)vi * addvec_ld;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int chunk = lane + k * 64;
                    if (chunk < chunks) { a0[k] = *(const float4*)(av + chunk * 8); a1[k] = *(const float4*)(av + chunk * 8 + 4); }
                }
                vi_loaded = vi;
            }
            s[r] = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int chunk = lane + k * 64;
                if (chunk < chunks) {
                    raw8_to_f<F32IN>(raw[r][k], v[r][k]);
                    if (addvec) {
                        v[r][k][0] += a0[k].x; v[r][k][1] += a0[k].y; v[r][k][2] += a0[k].z; v[r][k][3] += a0[k].w;
                        v[r][k][4] += a1[k].x; v[r][k][5] += a1[k].y; v[r][k][6] += a1[k].z; v[r][k][7] += a1[k].w;
                        if (sum_out) {
                            // the stored sum is bf16 (the residual stream dtype); normalise what is stored
                            const uint4 pk = pack8(v[r][k]);
                            if (live) *(uint4*)(sum_out + row * C + chunk * 8) = pk;
                            unpack8(pk, v[r][k]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[r] += v[r][k][e];
                }
            }
        }
        float mean[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) * inv_c;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            q[r] = 0.f;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int chunk = lane + k * 64;
                if (chunk < chunks) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[r][k][e] - mean[r]; q[r] += d * d; }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = rsqrtf(wave_sum(q[r]) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int chunk = lane + k * 64;
            if (chunk < chunks) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t row = rowg + r;
                    const float m = mean[r], rs = q[r];
                    float o[8];
                    o[0] = (v[r][k][0] - m) * rs * g0[k].x + b0[k].x; o[1] = (v[r][k][1] - m) * rs * g0[k].y + b0[k].y;
                    o[2] = (v[r][k][2] - m) * rs * g0[k].z + b0[k].z; o[3] = (v[r][k][3] - m) * rs * g0[k].w + b0[k].w;
                    o[4] = (v[r][k][4] - m) * rs * g1[k].x + b1[k].x; o[5] = (v[r][k][5] - m) * rs * g1[k].y + b1[k].y;
                    o[6] = (v[r][k][6] - m) * rs * g1[k].z + b1[k].z; o[7] = (v[r][k][7] - m) * rs * g1[k].w + b1[k].w;
                    if (row < run1) *(uint4*)(out + row * C + chunk * 8) = pack8(o);
                }
            }
        }
      }
    }
}

inline int grid_for(int64_t total, int block, int cap) {
    int64_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

namespace {
// Rows per block are chosen by the caller (or default to a function of the unit size) and NEVER depend on the number
// of units: the summation order of a unit — and with it every bit of a candidate's result — is then the same whether
// the candidate is evaluated alone or in a batch.
inline void gn_stats_geometry(int64_t rows, int rows_per_unit, int rows_per_block_in, int* units, int* splits, int* rows_per_block) {
    *units = (int)(rows / rows_per_unit);
    *rows_per_block = rows_per_block_in > 0 ? rows_per_block_in : (rows_per_unit <= 16384 ? 128 : 256);
    *splits = (rows_per_unit + *rows_per_block - 1) / *rows_per_block;
}
}  // namespace

extern "C" int64_t wiw_groupnorm_scratch_floats(int64_t rows, int rows_per_unit, int rows_per_block) {
    if (rows <= 0 || rows_per_unit <= 0 || rows % rows_per_unit != 0 || rows_per_block < 0) return 0;
    int units, splits, rpb;
    gn_stats_geometry(rows, rows_per_unit, rows_per_block, &units, &splits, &rpb);
    const int nparts = (splits + GN_PART - 1) / GN_PART;
    return (int64_t)units * splits * (GROUPS * 2) + (int64_t)units * nparts * (GROUPS * 4);
}

extern "C" int64_t wiw_groupnorm_counters(int64_t rows, int rows_per_unit, int rows_per_block) {
    if (rows <= 0 || rows_per_unit <= 0 || rows % rows_per_unit != 0 || rows_per_block < 0) return 0;
    int units, splits, rpb;
    gn_stats_geometry(rows, rows_per_unit, rows_per_block, &units, &splits, &rpb);
    return (int64_t)units * ((splits + GN_PART - 1) / GN_PART + 1);
}

namespace {
template <bool F32IN>
int gn_stats_launch(const char* what, void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                    int rows_per_unit, int rows_per_block_in, float* stats, float* scratch, unsigned* counters) {
    WIW_REQUIRE(X1 && stats && scratch && counters, "groupnorm_stats: null pointer");
    WIW_REQUIRE(rows_per_block_in >= 0, "groupnorm_stats: rows_per_block must be >= 0 (0 = default)");
    WIW_REQUIRE((C2 == 0) == (X2 == nullptr), "groupnorm_stats: X2 iff C2 > 0");
    const int C = C1 + C2;
    WIW_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && C % GROUPS == 0 && C <= GN_MAXC,
                "groupnorm_stats: channels must be %8, C %32 and C <= 4096");
    WIW_REQUIRE(rows > 0 && rows_per_unit > 0 && rows % rows_per_unit == 0, "groupnorm_stats: bad rows");
    int units, splits, rows_per_block;
    gn_stats_geometry(rows, rows_per_unit, rows_per_block_in, &units, &splits, &rows_per_block);
    hipLaunchKernelGGL(gn_stats_kernel<F32IN>, dim3(splits, units), dim3(256), (size_t)C * 2 * sizeof(float),
                       (hipStream_t)stream, X1, C1, X2, C2, rows_per_unit, rows_per_block, scratch, counters, stats);
    return wiw_check_launch(what);
}
}  // namespace

extern "C" int wiw_groupnorm_stats(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                                   int rows_per_unit, int rows_per_block_in, float* stats, float* scratch, unsigned* counters) {
    return gn_stats_launch<false>("wiw_groupnorm_stats", stream, X1, C1, X2, C2, rows, rows_per_unit, rows_per_block_in, stats,
                                  scratch, counters);
}

extern "C" int wiw_groupnorm_stats_f32in(void* stream, const float* X1, int C1, const float* X2, int C2, int64_t rows,
                                         int rows_per_unit, int rows_per_block_in, float* stats, float* scratch,
                                         unsigned* counters) {
    return gn_stats_launch<true>("wiw_groupnorm_stats_f32in", stream, X1, C1, X2, C2, rows, rows_per_unit, rows_per_block_in,
                                 stats, scratch, counters);
}

extern "C" int wiw_groupnorm_finalize(void* stream, const float* stats, const float* gamma, const float* beta,
                                      int units, int C, int rows_per_unit, float eps, float* ab) {
    WIW_REQUIRE(stats && gamma && beta && ab, "groupnorm_finalize: null pointer");
    WIW_REQUIRE(units > 0 && C > 0 && C % GROUPS == 0 && rows_per_unit > 0, "groupnorm_finalize: bad sizes");
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / GROUPS));
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(grid_for((int64_t)units * C, 256, 1024)), dim3(256), 0,
                       (hipStream_t)stream, stats, gamma, beta, units, C, inv_count, eps, ab);
    return wiw_check_launch("wiw_groupnorm_finalize");
}

extern "C" int wiw_groupnorm_apply(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                                   int rows_per_unit, const float* ab, int silu, void* out) {
    WIW_REQUIRE(X1 && ab && out, "groupnorm_apply: null pointer");
    WIW_REQUIRE((C2 == 0) == (X2 == nullptr), "groupnorm_apply: X2 iff C2 > 0");
    WIW_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0, "groupnorm_apply: channels must be multiples of 8");
    WIW_REQUIRE(rows > 0 && rows_per_unit > 0 && rows % rows_per_unit == 0, "groupnorm_apply: bad rows");
    const int units = (int)(rows / rows_per_unit);
    int splits = (4096 + units - 1) / units;
    const int max_splits = (rows_per_unit + 15) / 16;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int rows_per_block = (rows_per_unit + splits - 1) / splits;
    splits = (rows_per_unit + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((gn_apply_kernel<false, false>), dim3(splits, units), dim3(256), 0, (hipStream_t)stream, X1, C1, X2, C2,
                       rows_per_unit, rows_per_block, ab, silu, (uint16_t*)out, nullptr, nullptr, 0.f, 0.f, nullptr);
    return wiw_check_launch("wiw_groupnorm_apply");
}

namespace {
template <bool F32IN>
int gn_apply_stats_launch(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows, int rows_per_unit,
                          const float* stats, const float* gamma, const float* beta, float eps, int silu, void* out, void* raw16) {
    WIW_REQUIRE(X1 && stats && gamma && beta && out, "groupnorm_apply_stats: null pointer");
    WIW_REQUIRE((C2 == 0) == (X2 == nullptr), "groupnorm_apply_stats: X2 iff C2 > 0");
    const int C = C1 + C2;
    WIW_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && C % GROUPS == 0, "groupnorm_apply_stats: channels must be %8 and C %32");
    WIW_REQUIRE(rows > 0 && rows_per_unit > 0 && rows % rows_per_unit == 0, "groupnorm_apply_stats: bad rows");
    const int units = (int)(rows / rows_per_unit);
    int splits = (4096 + units - 1) / units;
    const int max_splits = (rows_per_unit + 15) / 16;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int rows_per_block = (rows_per_unit + splits - 1) / splits;
    splits = (rows_per_unit + rows_per_block - 1) / rows_per_block;
    const float inv_count = 1.0f / ((float)rows_per_unit * (float)(C / GROUPS));
    hipLaunchKernelGGL((gn_apply_kernel<true, F32IN>), dim3(splits, units), dim3(256), 0, (hipStream_t)stream, X1, C1, X2, C2,
                       rows_per_unit, rows_per_block, stats, silu, (uint16_t*)out, gamma, beta, inv_count, eps, (uint16_t*)raw16);
    return wiw_check_launch("wiw_groupnorm_apply_stats");
}
}  // namespace

extern "C" int wiw_groupnorm_apply_stats(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                                         int rows_per_unit, const float* stats, const float* gamma, const float* beta,
                                         float eps, int silu, void* out) {
    return gn_apply_stats_launch<false>(stream, X1, C1, X2, C2, rows, rows_per_unit, stats, gamma, beta, eps, silu, out, nullptr);
}

extern "C" int wiw_groupnorm_apply_stats_f32in(void* stream, const float* X1, int C1, const float* X2, int C2, int64_t rows,
                                               int rows_per_unit, const float* stats, const float* gamma, const float* beta,
                                               float eps, int silu, void* out, void* raw16) {
    return gn_apply_stats_launch<true>(stream, X1, C1, X2, C2, rows, rows_per_unit, stats, gamma, beta, eps, silu, out, raw16);
}

// One-pass GroupNorm(32)(+SiLU) of the small levels (ABI 15): rows_per_unit in {144, 576}, 40 or 80 channels per group,
// C1 (and C2) multiples of 160, 16-bit tensors.  wiw_groupnorm_onepass_ok says whether a launch fits.
extern "C" int wiw_groupnorm_onepass_ok(int C1, int C2, int64_t rows, int rows_per_unit) {
    const int C = C1 + C2;
    if (rows_per_unit != 144 && rows_per_unit != 576) return 0;
    if (rows <= 0 || rows % rows_per_unit) return 0;
    if (C % GROUPS || ((C / GROUPS) != 40 && (C / GROUPS) != 80)) return 0;
    if (C1 % GN1_CB || C2 % GN1_CB || C > GN_MAXC) return 0;
    return 1;
}
extern "C" int wiw_groupnorm_onepass(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows, int rows_per_unit,
                                     const float* gamma, const float* beta, float eps, int silu, void* out) {
    WIW_REQUIRE(wiw_groupnorm_onepass_ok(C1, C2, rows, rows_per_unit), "groupnorm_onepass: shape outside the one-pass kernel (wiw_groupnorm_onepass_ok)");
    WIW_REQUIRE(X1 && gamma && beta && out && (C2 == 0 || X2), "groupnorm_onepass: null pointer");
    const dim3 grid((unsigned)((C1 + C2) / GN1_CB), (unsigned)(rows / rows_per_unit));
    if (rows_per_unit == 576)
        hipLaunchKernelGGL(gn_onepass_kernel<48>, grid, dim3(GN1_THREADS), 0, (hipStream_t)stream, (const uint16_t*)X1, C1, (const uint16_t*)X2, C2,
                           gamma, beta, eps, silu, (uint16_t*)out);
    else
        hipLaunchKernelGGL(gn_onepass_kernel<12>, grid, dim3(GN1_THREADS), 0, (hipStream_t)stream, (const uint16_t*)X1, C1, (const uint16_t*)X2, C2,
                           gamma, beta, eps, silu, (uint16_t*)out);
    return wiw_check_launch("wiw_groupnorm_onepass");
}

extern "C" int wiw_layernorm_bf16(void* stream, const void* X, int64_t rows, int C, const float* gamma,
                                  const float* beta, float eps, const float* addvec, int addvec_ld, int rows_per_vec,
                                  void* sum_out, void* out) {
    WIW_REQUIRE(X && gamma && beta && out, "layernorm: null pointer");
    WIW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= LN_MAXCH * 64 * 8, "layernorm: C must be %8 and <= 2048");
    WIW_REQUIRE(addvec == nullptr || (rows_per_vec > 0 && addvec_ld % 4 == 0), "layernorm: bad addvec layout");
    WIW_REQUIRE(sum_out == nullptr || addvec != nullptr, "layernorm: sum_out requires addvec");
    const dim3 grid(grid_for(rows, 4 * LN_RUN, 256 * 8));   // <= 8 blocks per CU, grid-stride over runs of LN_RUN rows
#define WIW_LN_LAUNCH(NCH, R)                                                                                        \
    hipLaunchKernelGGL((layernorm_kernel<NCH, R, false>), grid, dim3(256), 0, (hipStream_t)stream, X, rows, C,        \
                       gamma, beta, eps, addvec, addvec_ld, rows_per_vec, (uint16_t*)sum_out, (uint16_t*)out)
    if (C <= 512) WIW_LN_LAUNCH(1, 8);
    else if (C <= 1024) WIW_LN_LAUNCH(2, LN_R_MID);
    else WIW_LN_LAUNCH(4, LN_R_WIDE);
#undef WIW_LN_LAUNCH
    return wiw_check_launch("wiw_layernorm_bf16");
}

extern "C" int wiw_layernorm_f32in(void* stream, const float* X, int64_t rows, int C, const float* gamma, const float* beta,
                                   float eps, void* out) {
    WIW_REQUIRE(X && gamma && beta && out, "layernorm_f32in: null pointer");
    WIW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= LN_MAXCH * 64 * 8, "layernorm_f32in: C must be %8 and <= 2048");
    const dim3 grid(grid_for(rows, 4 * LN_RUN, 256 * 8));
    // fp32 rows are twice as long: half as many rows in flight per wave keep the register budget of the 16-bit kernels
#define WIW_LN_LAUNCH(NCH, R)                                                                                             \
    hipLaunchKernelGGL((layernorm_kernel<NCH, R, true>), grid, dim3(256), 0, (hipStream_t)stream, (const void*)X, rows, C, \
                       gamma, beta, eps, (const float*)nullptr, 0, 1, (uint16_t*)nullptr, (uint16_t*)out)
    if (C <= 512) WIW_LN_LAUNCH(1, 4);
    else if (C <= 1024) WIW_LN_LAUNCH(2, 2);
    else WIW_LN_LAUNCH(4, 1);
#undef WIW_LN_LAUNCH
    return wiw_check_launch("wiw_layernorm_f32in");
}
