// Kernels of the fine-tuning step (FTsvd/train_svd.py:844-970): the AdamW update, the
// EDM loss with its gradient, and the backward kernels of the UNet operators that are not GEMM-shaped — attention
// (dQ / dK / dV, LDS-tiled), GroupNorm(+SiLU), LayerNorm, GEGLU — plus the weight-gradient kernel on row-major operands
// (wiw_wgrad_tn_bf16), column sums, the transposed im2col gather and the small element-wise helpers.  (Data gradients of the
// linear / convolution layers are wiw_gemm_bf16 launches on re-laid-out weights.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace {

// torch.optim.AdamW (decoupled weight decay), one fused pass over a flat parameter range, fp32 master weights:
//   p <- p * (1 - lr * wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;
//   p <- p - lr * (m / bc1) / (sqrt(v / bc2) + eps)          bc1 = 1 - b1^step, bc2 = 1 - b2^step
// and the 16-bit copy the forward / backward GEMMs read is refreshed in the same pass (24 B of HBM traffic per
// parameter + 2 B for the copy: the update of the 1.53 B-parameter UNet is ~40 GB, HBM-bound).
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, uint16_t* __restrict__ p16, int64_t n, float lr,
                                                     float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pp = *(const float4*)(p + i), mm = *(const float4*)(m + i), vv = *(const float4*)(v + i);
            const float4 gg = *(const float4*)(g + i);
            float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
            const float ga[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pa[e] *= 1.0f - lr * wd;
                ma[e] = b1 * ma[e] + (1.0f - b1) * ga[e];
                va[e] = b2 * va[e] + (1.0f - b2) * ga[e] * ga[e];
                pa[e] -= lr * (ma[e] * inv_bc1) / (sqrtf(va[e]) * inv_sqrt_bc2 + eps);
            }
            *(float4*)(p + i) = float4{pa[0], pa[1], pa[2], pa[3]};
            *(float4*)(m + i) = float4{ma[0], ma[1], ma[2], ma[3]};
            *(float4*)(v + i) = float4{va[0], va[1], va[2], va[3]};
            if (p16) {
                uint2 pk;
                pk.x = pack2bf(pa[0], pa[1]);
                pk.y = pack2bf(pa[2], pa[3]);
                *(uint2*)(p16 + i) = pk;
            }
        } else {
            for (int64_t j = i; j < n; ++j) {
                float pj = p[j] * (1.0f - lr * wd);
                const float mj = b1 * m[j] + (1.0f - b1) * g[j];
                const float vj = b2 * v[j] + (1.0f - b2) * g[j] * g[j];
                pj -= lr * (mj * inv_bc1) / (sqrtf(vj) * inv_sqrt_bc2 + eps);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (p16) p16[j] = f2bf(pj);
            }
        }
    }
}

// EDM loss of one sample and its gradient w.r.t. the model prediction (train_svd.py:940-952):
//   denoised = v * c_out + c_skip * noisy;  loss = mean(w * (denoised - target)^2);  dL/dv = 2 w c_out (denoised - target) / n
// partial[block] = sum of the block's terms (fixed order: deterministic); grad: fp32.
__global__ __launch_bounds__(256) void edm_loss_kernel(const float* __restrict__ pred, const float* __restrict__ noisy,
                                                        const float* __restrict__ target, int64_t n, float c_out, float c_skip,
                                                        float w, float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    const float scale = 2.0f * w * c_out / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = pred[i] * c_out + c_skip * noisy[i] - target[i];
        acc += w * d * d;
        grad[i] = scale * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward building blocks (16-bit activations / gradients, fp32 parameter gradients), all deterministic: per-block
// partials in a fixed order, no floating-point atomics.
// ---------------------------------------------------------------------------------------------------------------------

// out[p][c] = sum of the rows [p rpp, min(rows, (p + 1) rpp)) of X[rows][C], rpp = ceil(rows / P)  (X 16-bit or fp32, out fp32
// [P][C]): bias gradients (db = column sums of dY), per-frame sums (P = frames x k, then a second launch over the k partials
// of each frame) and the gamma / beta partials.  A workgroup sums a [rows of one part] x [32 sixteen-byte column chunks]
// panel: 8 row lanes x 32 chunk lanes, four independent 16-byte loads in flight per thread, the 8 row lanes combined through
// LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t rows, int C, int64_t rpp, float* __restrict__ out) {
    constexpr int CW = 16 / (int)sizeof(T);
    __shared__ float red[8][32 * CW + 4];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (blockIdx.x * 32 + cx) * CW;
    const int64_t r0 = (int64_t)blockIdx.y * rpp;
    const int64_t r1 = r0 + rpp < rows ? r0 + rpp : rows;
    float acc[4][CW];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[u][e] = 0.f;
    if (col < C) {
        const T* src = X + col;
        int64_t r = r0 + ry;
        for (; r + 24 < r1; r += 32) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(src + (r + 8 * u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                if constexpr (sizeof(T) == 2) unpack8(v[u], f);
                else { f[0] = __uint_as_float(v[u].x); f[1] = __uint_as_float(v[u].y); f[2] = __uint_as_float(v[u].z); f[3] = __uint_as_float(v[u].w); }
#pragma unroll
                for (int e = 0; e < CW; ++e) acc[u][e] += f[e];
            }
        }
        for (; r < r1; r += 8) {
            const uint4 v = *(const uint4*)(src + r * C);
            float f[8];
            if constexpr (sizeof(T) == 2) unpack8(v, f);
            else { f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w); }
#pragma unroll
            for (int e = 0; e < CW; ++e) acc[0][e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < CW; ++e) red[ry][cx * CW + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    __syncthreads();
    for (int t = threadIdx.x; t < 32 * CW; t += 256) {
        const int c = blockIdx.x * 32 * CW + t;
        if (c < C) {
            float sum = 0.f;
#pragma unroll
            for (int y = 0; y < 8; ++y) sum += red[y][t];
            out[(int64_t)blockIdx.y * C + c] = sum;
        }
    }
}

// LayerNorm backward, one wave per row (C <= 2048, C % 8 == 0), statistics recomputed from x (two-pass, fp32):
//   xhat = (x - mean) rstd;  g = dy gamma;  dx = rstd (g - mean_c(g) - xhat mean_c(g xhat))
// and per-WAVE partials of dgamma = sum_rows dy xhat, dbeta = sum_rows dy (fp32 [n_waves][2][C]; summed by colsum_kernel).
constexpr int LNB_MAXCH = 4;
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ dY,
                                                             const float* __restrict__ gamma, int64_t rows, int C, float eps,
                                                             const uint16_t* __restrict__ dRes, uint16_t* __restrict__ dX,
                                                             float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const int chunks = C >> 3;
    const float inv_c = 1.0f / (float)C;
    float dg[LNB_MAXCH][8], db[LNB_MAXCH][8], gm[LNB_MAXCH][8];
#pragma unroll
    for (int k = 0; k < LNB_MAXCH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dg[k][e] = 0.f; db[k][e] = 0.f;
            const int chunk = lane + k * 64;
            gm[k][e] = chunk < chunks ? gamma[chunk * 8 + e] : 0.f;
        }
    for (int64_t row = wave_g; row < rows; row += n_waves) {
        float x[LNB_MAXCH][8], dy[LNB_MAXCH][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LNB_MAXCH; ++k) {
            const int chunk = lane + k * 64;
            if (chunk < chunks) {
                unpack8(*(const uint4*)(X + row * C + chunk * 8), x[k]);
                unpack8(*(const uint4*)(dY + row * C + chunk * 8), dy[k]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[k][e] = 0.f; dy[k][e] = 0.f; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[k][e];
        }
        const float mean = wave_sum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LNB_MAXCH; ++k) {
            const bool on = lane + k * 64 < chunks;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = on ? x[k][e] - mean : 0.f; x[k][e] = d; q += d * d; }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LNB_MAXCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = x[k][e] * rstd;
                x[k][e] = xh;
                const float g = dy[k][e] * gm[k][e];
                s1 += g; s2 += g * xh;
                dg[k][e] += dy[k][e] * xh; db[k][e] += dy[k][e];
            }
        s1 = wave_sum(s1) * inv_c; s2 = wave_sum(s2) * inv_c;
#pragma unroll
        for (int k = 0; k < LNB_MAXCH; ++k) {
            const int chunk = lane + k * 64;
            if (chunk < chunks) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dy[k][e] * gm[k][e] - s1 - x[k][e] * s2);
                if (dRes) {   // gradient arriving through the residual connection around the normalised branch
                    float rr[8];
                    unpack8(*(const uint4*)(dRes + row * C + chunk * 8), rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rr[e];
                }
                *(uint4*)(dX + row * C + chunk * 8) = pack8(o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LNB_MAXCH; ++k) {
        const int chunk = lane + k * 64;
        if (chunk < chunks) {
            float* pg = partial + ((int64_t)wave_g * 2) * C + chunk * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pg[e] = dg[k][e]; pg[C + e] = db[k][e]; }
        }
    }
}

// GEGLU backward (activations.py:117-123): forward h = v * gelu_erf(g) with the projection output P = [v | g] ([rows][2 Ch]);
//   dP[:, :Ch] = dh * gelu(g);   dP[:, Ch:] = dh * v * (Phi(g) + g phi(g))
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint16_t* __restrict__ P, const uint16_t* __restrict__ dH,
                                                         int64_t rows, int Ch, uint16_t* __restrict__ dP) {
    const int64_t n = rows * (Ch >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (Ch >> 3);
        const int c = (int)(i - r * (Ch >> 3)) * 8;
        float v[8], g[8], dh[8], dv[8], dgt[8];
        unpack8(*(const uint4*)(P + r * 2 * Ch + c), v);
        unpack8(*(const uint4*)(P + r * 2 * Ch + Ch + c), g);
        unpack8(*(const uint4*)(dH + r * Ch + c), dh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ge = gelu_erf_as_f(g[e]);                                 // g * Phi(g)
            const float phi = 0.3989422804014327f * __expf(-0.5f * g[e] * g[e]);
            const float Phi = fabsf(g[e]) > 1e-6f ? ge / g[e] : 0.5f;
            dv[e] = dh[e] * ge;
            dgt[e] = dh[e] * v[e] * (Phi + g[e] * phi);
        }
        *(uint4*)(dP + r * 2 * Ch + c) = pack8(dv);
        *(uint4*)(dP + r * 2 * Ch + Ch + c) = pack8(dgt);
    }
}

// GroupNorm(32) (+ SiLU) backward on token-major [rows][C] tensors, units of `rows_per_unit` rows (a frame, or a clip for the
// temporal blocks), statistics (mean, biased variance) per (unit, group) from wiw_groupnorm_stats:
//   xhat = (x - mean_g) rstd_g;  z = xhat gamma_c + beta_c;  y = silu(z) | z;  dz = dy silu'(z) | dy
//   dgamma_c = sum dz xhat, dbeta_c = sum dz;  per (unit, group): A = sum dz gamma, B = sum dz gamma xhat, n = rows cg
//   dx = rstd_g (dz gamma_c - A / n - xhat B / n)
// Pass 1 (gnb_reduce): per (unit, row split) block partials of sum dz and sum dz xhat per channel, fixed order.
// Pass 2 (gnb_finish): sums the splits of a unit in split order -> unit_cs[unit][2][C] and AB[unit][32][2].
// Pass 3 (gnb_apply): elementwise dx.
constexpr int GNB_GROUPS = 32;
WIW_DEV float gnb_dz(float dy, float z, int silu) {
    if (!silu) return dy;
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    return dy * sg * (1.0f + z * (1.0f - sg));
}

// Per-thread constants of one 8-channel chunk of one unit: everything the row loops need, loaded once.
struct GnbChunk {
    float mean[8], rstd[8], gm[8], bt[8];
};
WIW_DEV void gnb_load_chunk(GnbChunk& k, const float* __restrict__ stats, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int unit, int c0, int cg, float eps) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c0 + e) / cg;
        k.mean[e] = stats[((int64_t)unit * GNB_GROUPS + g) * 2];
        k.rstd[e] = rsqrtf(stats[((int64_t)unit * GNB_GROUPS + g) * 2 + 1] + eps);
        k.gm[e] = gamma[c0 + e]; k.bt[e] = beta[c0 + e];
    }
}

// Blocks (row split, unit); threads = (8-channel chunk, row lane); two rows in flight per thread.
__global__ __launch_bounds__(256) void gnb_reduce_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ dY,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C, int rows_per_unit,
                                                          int rows_per_block, float eps, int silu, float* __restrict__ partial) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
    const int chunks = C >> 3, cg = C / GNB_GROUPS;
    const int cpb = chunks < 256 ? chunks : 256, rp = 256 / cpb;
    const int ci = tid % cpb, rl = tid / cpb;
    const int unit = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < rows_per_unit ? r0 + rows_per_block : rows_per_unit;
    const int64_t base = (int64_t)unit * rows_per_unit;
    for (int cbase = 0; cbase < chunks; cbase += cpb) {
        const int chunk = cbase + ci;
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
        if (rl < rp && chunk < chunks) {
            const int c0 = chunk * 8;
            GnbChunk k;
            gnb_load_chunk(k, stats, gamma, beta, unit, c0, cg, eps);
            for (int r = r0 + rl; r < r1; r += 2 * rp) {
                const bool two = r + rp < r1;
                const uint4 xa = *(const uint4*)(X + (base + r) * C + c0), da = *(const uint4*)(dY + (base + r) * C + c0);
                uint4 xb = xa, db = uint4{0u, 0u, 0u, 0u};
                if (two) { xb = *(const uint4*)(X + (base + r + rp) * C + c0); db = *(const uint4*)(dY + (base + r + rp) * C + c0); }
                float x[8], dy[8], x2[8], dy2[8];
                unpack8(xa, x); unpack8(da, dy); unpack8(xb, x2); unpack8(db, dy2);      // a zero dy row adds exactly 0
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (x[e] - k.mean[e]) * k.rstd[e], xh2 = (x2[e] - k.mean[e]) * k.rstd[e];
                    const float dz = gnb_dz(dy[e], __builtin_fmaf(xh, k.gm[e], k.bt[e]), silu);
                    const float dz2 = gnb_dz(dy2[e], __builtin_fmaf(xh2, k.gm[e], k.bt[e]), silu);
                    a[e] += dz; b[e] = __builtin_fmaf(dz, xh, b[e]);
                    a[e] += dz2; b[e] = __builtin_fmaf(dz2, xh2, b[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid][e] = a[e]; red[tid][8 + e] = b[e]; }
        __syncthreads();
        for (int t = tid; t < cpb * 16; t += 256) {       // (chunk column, value index): row lanes summed in lane order
            const int cc = t >> 4, e = t & 15;
            if (cbase + cc < chunks) {
                float sum = 0.f;
                for (int j = 0; j < rp; ++j) sum += red[j * cpb + cc][e];
                float* dst = partial + (((int64_t)unit * gridDim.x + blockIdx.x) * 2 + (e >> 3)) * C + (cbase + cc) * 8 + (e & 7);
                *dst = sum;
            }
        }
        __syncthreads();
    }
}

// Blocks (unit, group): the splits of the group's channels summed by 256 / cg split lanes, the lanes combined in lane order.
__global__ __launch_bounds__(256) void gnb_finish_kernel(const float* __restrict__ partial, int splits, int C,
                                                          const float* __restrict__ gamma, float* __restrict__ unit_cs,
                                                          float* __restrict__ AB) {
    __shared__ float ra[256], rb[256], sA[128], sB[128];
    const int unit = blockIdx.x, g = blockIdx.y, cg = C / GNB_GROUPS;      // cg <= 128
    const int lanes = 256 / cg, cl = threadIdx.x % cg, sl = threadIdx.x / cg;
    const int c = g * cg + cl;
    float a = 0.f, b = 0.f;
    if (sl < lanes)
        for (int sidx = sl; sidx < splits; sidx += lanes) {
            a += partial[(((int64_t)unit * splits + sidx) * 2) * C + c];
            b += partial[(((int64_t)unit * splits + sidx) * 2 + 1) * C + c];
        }
    ra[threadIdx.x] = a; rb[threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < cg) {
        a = 0.f; b = 0.f;
        for (int j = 0; j < lanes; ++j) { a += ra[j * cg + threadIdx.x]; b += rb[j * cg + threadIdx.x]; }
        unit_cs[((int64_t)unit * 2) * C + c] = a;
        unit_cs[((int64_t)unit * 2 + 1) * C + c] = b;
        sA[threadIdx.x] = a * gamma[c]; sB[threadIdx.x] = b * gamma[c];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0.f; b = 0.f;
        for (int j = 0; j < cg; ++j) { a += sA[j]; b += sB[j]; }
        AB[((int64_t)unit * GNB_GROUPS + g) * 2] = a;
        AB[((int64_t)unit * GNB_GROUPS + g) * 2 + 1] = b;
    }
}

// Same block / thread map as gnb_reduce: the per-chunk constants are loaded once per thread, two rows in flight.
__global__ __launch_bounds__(256) void gnb_apply_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ dY,
                                                         const float* __restrict__ stats, const float* __restrict__ AB,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int C, int rows_per_unit, int rows_per_block, float eps, int silu,
                                                         uint16_t* __restrict__ dX) {
    const int tid = threadIdx.x;
    const int chunks = C >> 3, cg = C / GNB_GROUPS;
    const int cpb = chunks < 256 ? chunks : 256, rp = 256 / cpb;
    const int ci = tid % cpb, rl = tid / cpb;
    const int unit = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = r0 + rows_per_block < rows_per_unit ? r0 + rows_per_block : rows_per_unit;
    const int64_t base = (int64_t)unit * rows_per_unit;
    const float inv_n = 1.0f / ((float)rows_per_unit * (float)cg);
    if (rl >= rp) return;
    for (int chunk = ci; chunk < chunks; chunk += cpb) {
        const int c0 = chunk * 8;
        GnbChunk k;
        gnb_load_chunk(k, stats, gamma, beta, unit, c0, cg, eps);
        float A[8], B[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c0 + e) / cg;
            A[e] = AB[((int64_t)unit * GNB_GROUPS + g) * 2] * inv_n;
            B[e] = AB[((int64_t)unit * GNB_GROUPS + g) * 2 + 1] * inv_n;
        }
        for (int r = r0 + rl; r < r1; r += 2 * rp) {
            const bool two = r + rp < r1;
            const uint4 xa = *(const uint4*)(X + (base + r) * C + c0), da = *(const uint4*)(dY + (base + r) * C + c0);
            uint4 xb = xa, db = da;
            if (two) { xb = *(const uint4*)(X + (base + r + rp) * C + c0); db = *(const uint4*)(dY + (base + r + rp) * C + c0); }
            float x[8], dy[8], x2[8], dy2[8], o[8], o2[8];
            unpack8(xa, x); unpack8(da, dy); unpack8(xb, x2); unpack8(db, dy2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (x[e] - k.mean[e]) * k.rstd[e], xh2 = (x2[e] - k.mean[e]) * k.rstd[e];
                const float dz = gnb_dz(dy[e], __builtin_fmaf(xh, k.gm[e], k.bt[e]), silu);
                const float dz2 = gnb_dz(dy2[e], __builtin_fmaf(xh2, k.gm[e], k.bt[e]), silu);
                o[e] = k.rstd[e] * (dz * k.gm[e] - A[e] - xh * B[e]);
                o2[e] = k.rstd[e] * (dz2 * k.gm[e] - A[e] - xh2 * B[e]);
            }
            *(uint4*)(dX + (base + r) * C + c0) = pack8(o);
            if (two) *(uint4*)(dX + (base + r + rp) * C + c0) = pack8(o2);
        }
    }
}

// im2col rows for the weight gradient of the implicit-GEMM convolutions: Xcol[m][tap * C + c] = X[src(m, tap)][c] (zeros
// outside the image / clip), taps = 9 (3x3, pad 1 over (H, W)) or 3 ((3,1,1) over T).  dW = dY^T . Xcol is then ONE GEMM
// whose K loop runs over the M rows (world-in-world_amd/train.py); 16 bytes per thread and tap.
__global__ __launch_bounds__(256) void gather_taps_kernel(const uint16_t* __restrict__ X, int64_t M, int C, int H, int W, int T,
                                                           int temporal, int stride, uint16_t* __restrict__ Xcol) {
    const int chunks = C >> 3, taps = temporal ? 3 : 9;
    const int HW = H * W;
    const int64_t total = M * taps * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % chunks);
        const int64_t mt = i / chunks;
        const int tap = (int)(mt % taps);
        const int64_t m = mt / taps;
        int64_t src = -1;
        if (temporal) {
            const int t = (int)((m / HW) % T) + tap - 1;
            if (t >= 0 && t < T) src = m + (int64_t)(tap - 1) * HW;
        } else {
            if (stride == 1) {
                const int rem = (int)(m % HW), y = rem / W + tap / 3 - 1, x = rem % W + tap % 3 - 1;
                if (y >= 0 && y < H && x >= 0 && x < W) src = m + (int64_t)(tap / 3 - 1) * W + (tap % 3 - 1);
            } else {   // stride 2, pad 1: output (H, W), input (2H, 2W)  (Downsample2D, downsampling.py:132-150)
                const int rem = (int)(m % HW), y = 2 * (rem / W) + tap / 3 - 1, x = 2 * (rem % W) + tap % 3 - 1;
                if (y >= 0 && y < 2 * H && x >= 0 && x < 2 * W) src = (m / HW) * (4 * HW) + (int64_t)y * (2 * W) + x;
            }
        }
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (src >= 0) v = *(const uint4*)(X + src * C + ch * 8);
        *(uint4*)(Xcol + (m * taps + tap) * C + ch * 8) = v;
    }
}

// The same im2col rows written TRANSPOSED: XcolT[tap * C + c][m] = X[src(m, tap)][c], m < M, zero columns for M <= m < Mp —
// directly the K-contiguous operand of the weight-gradient GEMM (K loop over the M rows), without materialising the
// [M][taps * C] rows and transposing them (9x the activation, written and read once more).  64 tokens x 64 channels per
// workgroup and tap through LDS; the 2 x 2 transposes in registers as in wiw_transpose_bf16 (elementwise.hip).
__global__ __launch_bounds__(256) void gather_taps_t_kernel(const uint16_t* __restrict__ X, int64_t M, int64_t Mp, int C, int H, int W,
                                                             int T, int temporal, int stride, uint16_t* __restrict__ XcolT) {
    __shared__ uint32_t tile[64][33];
    const int cblocks = (C + 63) >> 6;
    const int tap = blockIdx.y / cblocks, cb = (blockIdx.y % cblocks) * 64;
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int tid = threadIdx.x, HW = H * W;
    {
        const int p = tid >> 3, ch = tid & 7;
        uint4 v[2] = {uint4{0u, 0u, 0u, 0u}, uint4{0u, 0u, 0u, 0u}};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t m = m0 + 2 * p + q;
            int64_t src = -1;
            if (m < M && cb + ch * 8 < C) {
                if (temporal) {
                    const int t = (int)((m / HW) % T) + tap - 1;
                    if (t >= 0 && t < T) src = m + (int64_t)(tap - 1) * HW;
                } else if (stride == 1) {
                    const int rem = (int)(m % HW), y = rem / W + tap / 3 - 1, x = rem % W + tap % 3 - 1;
                    if (y >= 0 && y < H && x >= 0 && x < W) src = m + (int64_t)(tap / 3 - 1) * W + (tap % 3 - 1);
                } else {
                    const int rem = (int)(m % HW), y = 2 * (rem / W) + tap / 3 - 1, x = 2 * (rem % W) + tap % 3 - 1;
                    if (y >= 0 && y < 2 * H && x >= 0 && x < 2 * W) src = (m / HW) * (4 * HW) + (int64_t)y * (2 * W) + x;
                }
            }
            if (src >= 0) v[q] = *(const uint4*)(X + src * C + cb + ch * 8);
        }
        const uint32_t au[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, bu[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[ch * 8 + 2 * e][p] = (au[e] & 0xffffu) | (bu[e] << 16);
            tile[ch * 8 + 2 * e + 1][p] = (au[e] >> 16) | (bu[e] & 0xffff0000u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + it * 256;
        const int c = i >> 3, ch = i & 7;
        if (cb + c < C && m0 + ch * 8 < Mp)
            *(uint4*)(XcolT + ((int64_t)tap * C + cb + c) * Mp + m0 + ch * 8) =
                uint4{tile[c][ch * 4], tile[c][ch * 4 + 1], tile[c][ch * 4 + 2], tile[c][ch * 4 + 3]};
    }
}

// GEGLU forward on a SAVED projection output (training keeps P = [v | g] for the backward; inference fuses this into the
// GEMM epilogue on packed weights): H[r][c] = v * gelu_erf(g)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const uint16_t* __restrict__ P, int64_t rows, int Ch, uint16_t* __restrict__ Hh) {
    const int64_t n = rows * (Ch >> 3);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (Ch >> 3);
        const int c = (int)(i - r * (Ch >> 3)) * 8;
        float v[8], g[8];
        unpack8(*(const uint4*)(P + r * 2 * Ch + c), v);
        unpack8(*(const uint4*)(P + r * 2 * Ch + Ch + c), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_as_f(g[e]);
        *(uint4*)(Hh + r * Ch + c) = pack8(v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-attention backward (F.scaled_dot_product_attention of attention_processor.py:2383-2385), head_dim D, sequences of S
// rows at a row stride of Sp (S <= Sp, Sp % 16 == 0).  FIRST, simple form: one wave per (sequence, head, 16-row tile),
// operands straight from global memory, 16-deep contraction steps on the 16x16x32 MFMA (upper half of its K zero), the
// scores recomputed from Q, K and the saved row log-sum-exp.  Correct and deterministic; NOT tuned (every wave re-reads
// all keys / queries of its sequence through L2) — the LDS-tiled version follows the forward kernel's structure.
//   S = scale Q K^T, P = softmax(S), O = P V;   D_q = sum_d dO O
//   dS = P (dO V^T - D);   dQ = scale dS K;   dK = scale dS^T Q;   dV = P^T dO
// Kernel 1 (per query tile): row LSE (log2 domain) + D (written out for kernel 2) and dQ  — contracts over keys, needs K^T.
// Kernel 2 (per key tile): dK, dV — contracts over queries, needs Q^T and dO^T.  (wiw_transpose_bf16 makes the transposes.)
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const uint16_t* __restrict__ QKV, int ld, int k_off, int v_off,
                                                           const uint16_t* __restrict__ Kt, int64_t ldt,
                                                           const uint16_t* __restrict__ O, const uint16_t* __restrict__ dO, int ldo,
                                                           uint16_t* __restrict__ dQKV, int ldd, float* __restrict__ LSE,
                                                           float* __restrict__ Dsum, int S, int Sp, int heads, int q_tiles,
                                                           int64_t total, float scale, float scale_log2e) {
    constexpr int NKK = D / 32, NDB = D / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= total) return;
    const int qt = (int)(task % q_tiles);
    const int h = (int)((task / q_tiles) % heads);
    const int64_t seq = task / ((int64_t)q_tiles * heads);
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t row0 = seq * Sp;
    const int qi = qt * 16 + fr;                     // < Sp (Sp % 16 == 0)
    bf16x8 qf[NKK], dof[NKK];
    float dsum = 0.f;
    {
        const uint16_t* qs = QKV + (row0 + qi) * ld + h * D;
        const uint16_t* os = O + (row0 + qi) * ldo + h * D;
        const uint16_t* ds = dO + (row0 + qi) * ldo + h * D;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            qf[kk] = *(const bf16x8*)(qs + kk * 32 + fq * 8);
            dof[kk] = *(const bf16x8*)(ds + kk * 32 + fq * 8);
            float a[8], b[8];
            unpack8(*(const uint4*)(os + kk * 32 + fq * 8), a);
            unpack8(*(const uint4*)(ds + kk * 32 + fq * 8), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = __builtin_fmaf(a[e], b[e], dsum);
        }
    }
    dsum = xor32_sum(xor16_sum(dsum));               // D of query fr (the four fq lanes hold the four quarters of d)
    const int nkt = (S + 15) / 16;
    // ---- pass 1: row log-sum-exp (log2 domain)
    float m_run = -INFINITY, l_run = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        const uint16_t* ksrc = QKV + (row0 + kt * 16 + fr) * ld + k_off + h * D;
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) st = WIW_MFMA(*(const bf16x8*)(ksrc + kk * 32 + fq * 8), qf[kk], st);
        float mx = -INFINITY, sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv[r] = (kt * 16 + fq * 4 + r) < S ? st[r] * scale_log2e : -INFINITY; mx = fmaxf(mx, sv[r]); }
        mx = xor32_max(xor16_max(mx));
        const float m_new = fmaxf(m_run, mx);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) ls += __builtin_amdgcn_exp2f(sv[r] - m_new);
        ls = xor32_sum(xor16_sum(ls));
        l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + ls;
        m_run = m_new;
    }
    const float lse2 = m_run + __builtin_amdgcn_logf(l_run);     // log2(sum_k 2^(s_k))
    if (fq == 0) {
        LSE[(seq * heads + h) * Sp + qi] = lse2;
        Dsum[(seq * heads + h) * Sp + qi] = dsum;
    }
    // ---- pass 2: dQ^T[d][q] += K^T[d][key] dS^T[key][q]
    f32x4 dq[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
        const uint16_t* ksrc = QKV + (row0 + kt * 16 + fr) * ld + k_off + h * D;
        const uint16_t* vsrc = QKV + (row0 + kt * 16 + fr) * ld + v_off + h * D;
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            st = WIW_MFMA(*(const bf16x8*)(ksrc + kk * 32 + fq * 8), qf[kk], st);    // lane: query fr, keys 4 fq + r
            dp = WIW_MFMA(*(const bf16x8*)(vsrc + kk * 32 + fq * 8), dof[kk], dp);   // dP^T = V dO^T, same layout
        }
        float dsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = (kt * 16 + fq * 4 + r) < S;
            const float pr = ok ? __builtin_amdgcn_exp2f(st[r] * scale_log2e - lse2) : 0.f;
            dsv[r] = pr * (dp[r] - dsum);
        }
        union { uint32_t u[4]; bf16x8 v; } dso;
        dso.u[0] = pack2bf(dsv[0], dsv[1]); dso.u[1] = pack2bf(dsv[2], dsv[3]); dso.u[2] = 0u; dso.u[3] = 0u;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const uint2 kv = *(const uint2*)(Kt + (int64_t)(h * D + db * 16 + fr) * ldt + row0 + kt * 16 + fq * 4);
            union { uint32_t u[4]; bf16x8 v; } ko;
            ko.u[0] = kv.x; ko.u[1] = kv.y; ko.u[2] = 0u; ko.u[3] = 0u;
            dq[db] = WIW_MFMA(ko.v, dso.v, dq[db]);      // lane: query fr, d = 16 db + 4 fq + r
        }
    }
    uint16_t* dst = dQKV + (row0 + qi) * ldd + h * D + fq * 4;
    const float sc = qi < S ? scale : 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        uint2 pk;
        pk.x = pack2bf(dq[db][0] * sc, dq[db][1] * sc);
        pk.y = pack2bf(dq[db][2] * sc, dq[db][3] * sc);
        *(uint2*)(dst + db * 16) = pk;
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ QKV, int ld, int k_off, int v_off,
                                                            const uint16_t* __restrict__ Qt, const uint16_t* __restrict__ dOt,
                                                            int64_t ldt, const uint16_t* __restrict__ dO, int ldo,
                                                            uint16_t* __restrict__ dQKV, int ldd, const float* __restrict__ LSE,
                                                            const float* __restrict__ Dsum, int S, int Sp, int heads, int k_tiles,
                                                            int64_t total, float scale, float scale_log2e) {
    constexpr int NKK = D / 32, NDB = D / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= total) return;
    const int kt = (int)(task % k_tiles);
    const int h = (int)((task / k_tiles) % heads);
    const int64_t seq = task / ((int64_t)k_tiles * heads);
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t row0 = seq * Sp;
    const int ki = kt * 16 + fr;                      // this lane's key (B operand column)
    bf16x8 kf[NKK], vf[NKK];
    {
        const uint16_t* ks = QKV + (row0 + ki) * ld + k_off + h * D;
        const uint16_t* vs = QKV + (row0 + ki) * ld + v_off + h * D;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) { kf[kk] = *(const bf16x8*)(ks + kk * 32 + fq * 8); vf[kk] = *(const bf16x8*)(vs + kk * 32 + fq * 8); }
    }
    const bool key_ok = ki < S;
    f32x4 dk[NDB], dv[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = dk[db]; }
    const int nqt = (S + 15) / 16;
    for (int qt = 0; qt < nqt; ++qt) {
        const uint16_t* qsrc = QKV + (row0 + qt * 16 + fr) * ld + h * D;       // A operand rows = queries
        const uint16_t* dsrc = dO + (row0 + qt * 16 + fr) * ldo + h * D;
        f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            sm = WIW_MFMA(*(const bf16x8*)(qsrc + kk * 32 + fq * 8), kf[kk], sm);    // lane: key fr, queries 4 fq + r
            dp = WIW_MFMA(*(const bf16x8*)(dsrc + kk * 32 + fq * 8), vf[kk], dp);    // dP = dO V^T, same layout
        }
        const float4 l4 = *(const float4*)(LSE + (seq * heads + h) * Sp + qt * 16 + fq * 4);
        const float4 d4 = *(const float4*)(Dsum + (seq * heads + h) * Sp + qt * 16 + fq * 4);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
        float pv[4], dsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = key_ok && (qt * 16 + fq * 4 + r) < S;
            pv[r] = ok ? __builtin_amdgcn_exp2f(sm[r] * scale_log2e - lv[r]) : 0.f;
            dsv[r] = pv[r] * (dp[r] - dd[r]);
        }
        union { uint32_t u[4]; bf16x8 v; } po, dso;
        po.u[0] = pack2bf(pv[0], pv[1]); po.u[1] = pack2bf(pv[2], pv[3]); po.u[2] = 0u; po.u[3] = 0u;
        dso.u[0] = pack2bf(dsv[0], dsv[1]); dso.u[1] = pack2bf(dsv[2], dsv[3]); dso.u[2] = 0u; dso.u[3] = 0u;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            const int64_t trow = (int64_t)(h * D + db * 16 + fr) * ldt + row0 + qt * 16 + fq * 4;
            const uint2 a1 = *(const uint2*)(dOt + trow), a2 = *(const uint2*)(Qt + trow);
            union { uint32_t u[4]; bf16x8 v; } x1, x2;
            x1.u[0] = a1.x; x1.u[1] = a1.y; x1.u[2] = 0u; x1.u[3] = 0u;
            x2.u[0] = a2.x; x2.u[1] = a2.y; x2.u[2] = 0u; x2.u[3] = 0u;
            dv[db] = WIW_MFMA(x1.v, po.v, dv[db]);       // dV^T[d][key] += dO^T[d][q] P[q][key];  lane: key fr, d = 16 db + 4 fq + r
            dk[db] = WIW_MFMA(x2.v, dso.v, dk[db]);      // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
    }
    uint16_t* dkd = dQKV + (row0 + ki) * ldd + k_off + h * D + fq * 4;
    uint16_t* dvd = dQKV + (row0 + ki) * ldd + v_off + h * D + fq * 4;
    const float sc = key_ok ? scale : 0.f;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
        uint2 pk;
        pk.x = pack2bf(dk[db][0] * sc, dk[db][1] * sc); pk.y = pack2bf(dk[db][2] * sc, dk[db][3] * sc);
        *(uint2*)(dkd + db * 16) = pk;
        pk.x = pack2bf(dv[db][0], dv[db][1]); pk.y = pack2bf(dv[db][2], dv[db][3]);
        if (!key_ok) pk = uint2{0u, 0u};
        *(uint2*)(dvd + db * 16) = pk;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-tiled forms of the two kernels above for the spatial attention's long sequences (S == Sp, S % 32 == 0, S >= 128).
// One workgroup = 4 waves = 128 rows of one (sequence, head); every wave keeps ITS 32 rows' fragments and accumulators in
// registers and the four share each 32-row tile of the other side through LDS (loaded once per workgroup, the next tile's
// global loads in flight behind the MFMAs) — the simple form re-read every tile per wave through L1, which bounded it.
// The contraction of the second GEMMs runs over 32 rows per MFMA (both halves of the 16x16x32 K used): the two score tiles
// of a step are packed into ONE B operand under the k permutation  e < 4 -> row 4 fq + e,  e >= 4 -> row 16 + 4 fq + e - 4,
// and the transposed A operand is read from the SAME row-major LDS tile under that permutation with two
// ds_read_b64_tr_b16 (atb_tr_frag) — Q^T, K^T, dO^T never exist in memory.
// Per step and wave: 32 MFMAs (dK/dV), 24 + 8 (dQ + its log-sum-exp pass), 16 KB of LDS reads.
// ---------------------------------------------------------------------------------------------------------------------
// Waves per SIMD the register allocator is held to (A/B knobs of the build: -DWIW_DKV_OCC= / -DWIW_DQ_OCC= lift them).  Both
// kernels are bound by VALU issue beside their MFMAs (exp2, the dS products, packing), so a third / fourth resident wave
// per SIMD is what fills the matrix pipe: dK/dV 164 VGPRs, dQ 124, no spills (profiles/r04b_attn_bwd_ab.txt).
#ifndef WIW_DKV_OCC
#define WIW_DKV_OCC __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#ifndef WIW_DQ_OCC
#define WIW_DQ_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
constexpr int ATB_ROW = 72;    // elements per LDS row of a [32 rows][64 d] tile (144 B: 16-byte reads of 16 rows spread over the banks)

// Transposed MFMA operand straight from a ROW-MAJOR [32 rows][64 d] LDS tile with ds_read_b64_tr_b16: every lane hands in
// the address of 4 contiguous elements, the 16 lanes of a group together one [4 rows][16 d] block (lane i: row i / 4,
// columns 4 (i % 4) ..), and lane i gets column i of the block (tools/ubench/tr_read.hip).  Two reads: rows 4 fq .. + 3
// and 16 + 4 fq .. + 3 of column d = 16 db + fr — the k permutation of the packed score operand.  No transposed copies
// of Q, K, dO in global memory, no second set of tiles.
typedef short atb_short4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 atb_tr_frag(const uint16_t* tile, int db, int fr, int fq) {
    const uint16_t* p = tile + (4 * fq + (fr >> 2)) * ATB_ROW + db * 16 + (fr & 3) * 4;
    union { atb_short4 h[2]; bf16x8 v; } x;
    x.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) atb_short4*)p);
    x.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) atb_short4*)(p + 16 * ATB_ROW));
    return x.v;
}

template <int D>
__global__ __launch_bounds__(256) WIW_DKV_OCC void attn_bwd_dkv_tiled_kernel(const uint16_t* __restrict__ QKV, int ld, int k_off, int v_off,
                                                                  const uint16_t* __restrict__ dO, int ldo,
                                                                  uint16_t* __restrict__ dQKV, int ldd, const float* __restrict__ LSE,
                                                                  const float* __restrict__ Dsum, int S, int heads, int k_blocks,
                                                                  float scale, float scale_log2e) {
    static_assert(D == 64, "head_dim 64");
    __shared__ __attribute__((aligned(16))) uint16_t Qs[32 * ATB_ROW];
    __shared__ __attribute__((aligned(16))) uint16_t dOs[32 * ATB_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int kb = (int)(blockIdx.x % k_blocks);
    const int h = (int)((blockIdx.x / k_blocks) % heads);
    const int64_t seq = blockIdx.x / ((int64_t)k_blocks * heads);
    const int64_t row0 = seq * S;
    const int key0 = kb * 128 + wave * 32;
    const bool active = key0 < S;                      // wave-uniform: S % 32 == 0
    bf16x8 kf[2][2], vf[2][2];
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int64_t krow = row0 + (active ? key0 + kt * 16 + fr : 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[kt][kk] = *(const bf16x8*)(QKV + krow * ld + k_off + h * D + kk * 32 + fq * 8);
            vf[kt][kk] = *(const bf16x8*)(QKV + krow * ld + v_off + h * D + kk * 32 + fq * 8);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) { dk[kt][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][db] = dk[kt][db]; }
    }
    const int lr = tid >> 3, lc = (tid & 7) * 8;       // loader: [32][64] tiles, 16 bytes per thread
    const uint16_t* gq = QKV + (row0 + lr) * ld + h * D + lc;
    const uint16_t* gdo = dO + (row0 + lr) * ldo + h * D + lc;
    uint4 r0 = *(const uint4*)gq, r1 = *(const uint4*)gdo;
    const float* lse_p = LSE + (seq * heads + h) * S;
    const float* dsum_p = Dsum + (seq * heads + h) * S;
    const int nq = S / 32;
    for (int qt = 0; qt < nq; ++qt) {
        __syncthreads();                               // every wave is done with the previous tile
        *(uint4*)(Qs + lr * ATB_ROW + lc) = r0;
        *(uint4*)(dOs + lr * ATB_ROW + lc) = r1;
        __syncthreads();
        if (qt + 1 < nq) {                             // next tile: in flight behind this tile's MFMAs
            r0 = *(const uint4*)(gq + (int64_t)(qt + 1) * 32 * ld);
            r1 = *(const uint4*)(gdo + (int64_t)(qt + 1) * 32 * ldo);
        }
        if (!active) continue;
        union { uint32_t u[4]; bf16x8 v; } po[2], dso[2];
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            bf16x8 aq[2], ado[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                aq[kk] = *(const bf16x8*)(Qs + (qs * 16 + fr) * ATB_ROW + kk * 32 + fq * 8);
                ado[kk] = *(const bf16x8*)(dOs + (qs * 16 + fr) * ATB_ROW + kk * 32 + fq * 8);
            }
            const float4 l4 = *(const float4*)(lse_p + qt * 32 + qs * 16 + fq * 4);
            const float4 d4 = *(const float4*)(dsum_p + qt * 32 + qs * 16 + fq * 4);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    sm = WIW_MFMA(aq[kk], kf[kt][kk], sm);       // lane: key fr of tile kt, queries 16 qs + 4 fq + r
                    dp = WIW_MFMA(ado[kk], vf[kt][kk], dp);
                }
                float pv[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = __builtin_amdgcn_exp2f(sm[r] * scale_log2e - lv[r]);
                    dsv[r] = pv[r] * (dp[r] - dd[r]);
                }
                po[kt].u[2 * qs] = pack2bf(pv[0], pv[1]);   po[kt].u[2 * qs + 1] = pack2bf(pv[2], pv[3]);
                dso[kt].u[2 * qs] = pack2bf(dsv[0], dsv[1]); dso[kt].u[2 * qs + 1] = pack2bf(dsv[2], dsv[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 x1 = atb_tr_frag(dOs, db, fr, fq);      // dO^T[d = 16 db + fr][queries 4 fq .., 16 + 4 fq ..]
            const bf16x8 x2 = atb_tr_frag(Qs, db, fr, fq);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                dv[kt][db] = WIW_MFMA(x1, po[kt].v, dv[kt][db]);     // dV^T[d][key] += dO^T[d][q] P[q][key], 32 queries deep
                dk[kt][db] = WIW_MFMA(x2, dso[kt].v, dk[kt][db]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        uint16_t* dkd = dQKV + (row0 + key0 + kt * 16 + fr) * ldd + k_off + h * D + fq * 4;   // lane: key fr, d = 16 db + 4 fq + r
        uint16_t* dvd = dQKV + (row0 + key0 + kt * 16 + fr) * ldd + v_off + h * D + fq * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 pk;
            pk.x = pack2bf(dk[kt][db][0] * scale, dk[kt][db][1] * scale); pk.y = pack2bf(dk[kt][db][2] * scale, dk[kt][db][3] * scale);
            *(uint2*)(dkd + db * 16) = pk;
            pk.x = pack2bf(dv[kt][db][0], dv[kt][db][1]); pk.y = pack2bf(dv[kt][db][2], dv[kt][db][3]);
            *(uint2*)(dvd + db * 16) = pk;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) WIW_DQ_OCC void attn_bwd_dq_tiled_kernel(const uint16_t* __restrict__ QKV, int ld, int k_off, int v_off,
                                                                 const uint16_t* __restrict__ O, const uint16_t* __restrict__ dO, int ldo,
                                                                 uint16_t* __restrict__ dQKV, int ldd, float* __restrict__ LSE,
                                                                 float* __restrict__ Dsum, int S, int heads, int q_blocks, float scale,
                                                                 float scale_log2e, int lse_given) {
    static_assert(D == 64, "head_dim 64");
    __shared__ __attribute__((aligned(16))) uint16_t Ks[32 * ATB_ROW];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[32 * ATB_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int qb = (int)(blockIdx.x % q_blocks);
    const int h = (int)((blockIdx.x / q_blocks) % heads);
    const int64_t seq = blockIdx.x / ((int64_t)q_blocks * heads);
    const int64_t row0 = seq * S;
    const int q0 = qb * 128 + wave * 32;
    const bool active = q0 < S;
    bf16x8 qf[2][2], dof[2][2];
    float dsum[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int64_t qrow = row0 + (active ? q0 + qt * 16 + fr : 0);
        const uint16_t* qs = QKV + qrow * ld + h * D;
        const uint16_t* os = O + qrow * ldo + h * D;
        const uint16_t* ds = dO + qrow * ldo + h * D;
        float acc = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[qt][kk] = *(const bf16x8*)(qs + kk * 32 + fq * 8);
            dof[qt][kk] = *(const bf16x8*)(ds + kk * 32 + fq * 8);
            float a[8], b[8];
            unpack8(*(const uint4*)(os + kk * 32 + fq * 8), a);
            unpack8(*(const uint4*)(ds + kk * 32 + fq * 8), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(a[e], b[e], acc);
        }
        dsum[qt] = xor32_sum(xor16_sum(acc));          // D of query fr of tile qt (same summation order as the simple form)
    }
    const int lr = tid >> 3, lc = (tid & 7) * 8;
    const uint16_t* gk = QKV + (row0 + lr) * ld + k_off + h * D + lc;
    const uint16_t* gv = QKV + (row0 + lr) * ld + v_off + h * D + lc;
    const int nk = S / 32;
    // ---- pass 1: row log-sum-exp (log2 domain) over all keys — skipped (block-uniformly) when the forward attention has
    // already written it (wiw_attn_spatial_lse_bf16): a whole Q.K^T pass less
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    uint4 r0 = *(const uint4*)gk, r1;
    for (int kt = 0; kt < (lse_given ? 0 : nk); ++kt) {
        __syncthreads();
        *(uint4*)(Ks + lr * ATB_ROW + lc) = r0;
        __syncthreads();
        if (kt + 1 < nk) r0 = *(const uint4*)(gk + (int64_t)(kt + 1) * 32 * ld);
        if (!active) continue;
        f32x4 st[2][2];                                // [key half ks][query tile qt]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ak[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) ak[kk] = *(const bf16x8*)(Ks + (ks * 16 + fr) * ATB_ROW + kk * 32 + fq * 8);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                st[ks][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) st[ks][qt] = WIW_MFMA(ak[kk], qf[qt][kk], st[ks][qt]);   // lane: query fr, keys 16 ks + 4 fq + r
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[ks][qt][r] * scale_log2e);
            mx = xor32_max(xor16_max(mx));
            const float m_new = fmaxf(m_run[qt], mx);
            float ls = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) ls += __builtin_amdgcn_exp2f(st[ks][qt][r] * scale_log2e - m_new);
            ls = xor32_sum(xor16_sum(ls));
            l_run[qt] = l_run[qt] * __builtin_amdgcn_exp2f(m_run[qt] - m_new) + ls;
            m_run[qt] = m_new;
        }
    }
    float lse2[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (lse_given) lse2[qt] = LSE[(seq * heads + h) * S + (active ? q0 + qt * 16 + fr : 0)];
        else lse2[qt] = m_run[qt] + __builtin_amdgcn_logf(l_run[qt]);
        if (active && fq == 0) {
            if (!lse_given) LSE[(seq * heads + h) * S + q0 + qt * 16 + fr] = lse2[qt];
            Dsum[(seq * heads + h) * S + q0 + qt * 16 + fr] = dsum[qt];
        }
    }
    // ---- pass 2: dQ^T[d][q] += K^T[d][key] dS^T[key][q], 32 keys deep
    f32x4 dq[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[qt][db] = f32x4{0.f, 0.f, 0.f, 0.f};
    r0 = *(const uint4*)gk; r1 = *(const uint4*)gv;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        *(uint4*)(Ks + lr * ATB_ROW + lc) = r0;
        *(uint4*)(Vs + lr * ATB_ROW + lc) = r1;
        __syncthreads();
        if (kt + 1 < nk) {
            r0 = *(const uint4*)(gk + (int64_t)(kt + 1) * 32 * ld);
            r1 = *(const uint4*)(gv + (int64_t)(kt + 1) * 32 * ld);
        }
        if (!active) continue;
        union { uint32_t u[4]; bf16x8 v; } dso[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ak[2], av[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                ak[kk] = *(const bf16x8*)(Ks + (ks * 16 + fr) * ATB_ROW + kk * 32 + fq * 8);
                av[kk] = *(const bf16x8*)(Vs + (ks * 16 + fr) * ATB_ROW + kk * 32 + fq * 8);
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    st = WIW_MFMA(ak[kk], qf[qt][kk], st);       // lane: query fr of tile qt, keys 16 ks + 4 fq + r
                    dp = WIW_MFMA(av[kk], dof[qt][kk], dp);      // dP^T = V dO^T
                }
                float dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dsv[r] = __builtin_amdgcn_exp2f(st[r] * scale_log2e - lse2[qt]) * (dp[r] - dsum[qt]);
                dso[qt].u[2 * ks] = pack2bf(dsv[0], dsv[1]);
                dso[qt].u[2 * ks + 1] = pack2bf(dsv[2], dsv[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 x = atb_tr_frag(Ks, db, fr, fq);        // K^T[d = 16 db + fr][keys 4 fq .., 16 + 4 fq ..]
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) dq[qt][db] = WIW_MFMA(x, dso[qt].v, dq[qt][db]);   // lane: query fr, d = 16 db + 4 fq + r
        }
    }
    if (!active) return;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        uint16_t* dst = dQKV + (row0 + q0 + qt * 16 + fr) * ldd + h * D + fq * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            uint2 pk;
            pk.x = pack2bf(dq[qt][db][0] * scale, dq[qt][db][1] * scale);
            pk.y = pack2bf(dq[qt][db][2] * scale, dq[qt][db][3] * scale);
            *(uint2*)(dst + db * 16) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient straight from ROW-MAJOR operands ("TN" GEMM):  dW[n][k] = sum_m dY[m][n] X[m][k]  (fp32), the contraction
// over the M rows of the layer.  wiw_gemm_bf16 wants both operands K-contiguous, i.e. dY^T and X^T made by two transposes
// per layer; here a workgroup stages [32 rows][256 n] of dY and [32 rows][128 k] of X row-major in LDS (64-column panels
// at the attention kernels' 144-byte row stride) and BOTH MFMA operands are read transposed with ds_read_b64_tr_b16
// (atb_tr_frag; the two reads per operand select rows 4 fq .. + 3 and 16 + 4 fq .. + 3 — the same permutation of the 32
// contracted rows on both sides, so the product is unchanged).  4 waves, wave w owns n = 64 w .. + 63 and all 128 k:
// 32 accumulator fragments, 24 transposed reads per 32 MFMAs; the next 32 rows' global loads are in flight behind them.
// The M rows are split over gridDim.y; split s writes its raw sums to slab s of `out` ([splits][N][K] fp32), summed in a
// fixed order by the caller (wiw_colsum over the slabs).  Rows >= M, columns >= N / K read as zeros.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WTN_BN = 256, WTN_BK = 128, WTN_PANEL = 32 * ATB_ROW;
__global__ __launch_bounds__(256) void wgrad_tn_kernel(const uint16_t* __restrict__ dY, int64_t ldy, const uint16_t* __restrict__ X,
                                                        int64_t ldx, int64_t M, int N, int K, int64_t rows_per_split,
                                                        float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t As[4 * WTN_PANEL];      // dY tile: 4 panels of [32 rows][64 n]
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2 * WTN_PANEL];      // X tile:  2 panels of [32 rows][64 k]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int kt_n = (K + WTN_BK - 1) / WTN_BK;
    const int n0 = (int)(blockIdx.x / kt_n) * WTN_BN, k0 = (int)(blockIdx.x % kt_n) * WTN_BK;
    const int64_t m_begin = (int64_t)blockIdx.y * rows_per_split;
    const int64_t m_end = m_begin + rows_per_split < M ? m_begin + rows_per_split : M;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // loader: A 32 rows x 32 chunks of 16 B (4 per thread), B 32 x 16 (2 per thread)
    const int a_row = tid >> 5, a_ch = tid & 31;          // rows a_row + 8 j
    const int b_row = tid >> 4, b_ch = tid & 15;          // rows b_row + 16 j
    const bool a_ok = n0 + a_ch * 8 < N, b_ok = k0 + b_ch * 8 < K;
    const uint16_t* ga = dY + n0 + a_ch * 8;
    const uint16_t* gb = X + k0 + b_ch * 8;
    uint16_t* sa = As + (a_ch >> 3) * WTN_PANEL + a_row * ATB_ROW + (a_ch & 7) * 8;
    uint16_t* sb = Bs + (b_ch >> 3) * WTN_PANEL + b_row * ATB_ROW + (b_ch & 7) * 8;
    uint4 ra[4], rb[2];
    auto fetch = [&](int64_t m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = m + a_row + 8 * j;
            ra[j] = (a_ok && r < m_end) ? *(const uint4*)(ga + r * ldy) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t r = m + b_row + 16 * j;
            rb[j] = (b_ok && r < m_end) ? *(const uint4*)(gb + r * ldx) : uint4{0u, 0u, 0u, 0u};
        }
    };
    fetch(m_begin);
    const uint16_t* a_panel = As + wave * WTN_PANEL;
    for (int64_t m = m_begin; m < m_end; m += 32) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint4*)(sa + 8 * j * ATB_ROW) = ra[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *(uint4*)(sb + 16 * j * ATB_ROW) = rb[j];
        __syncthreads();
        if (m + 32 < m_end) fetch(m + 32);
        bf16x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = atb_tr_frag(a_panel, i, fr, fq);          // dY^T[n = 64 wave + 16 i + fr][32 rows]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 bfr = atb_tr_frag(Bs + (j >> 2) * WTN_PANEL, j & 3, fr, fq);   // X[32 rows][k = 16 j + fr]
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = WIW_MFMA(af[i], bfr, acc[i][j]);
        }
    }
    // lane: n = n0 + 64 wave + 16 i + 4 fq + r, k = k0 + 16 j + fr
    float* slab = out + (int64_t)blockIdx.y * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wave * 64 + i * 16 + fq * 4 + r;
            if (n < N) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j * 16 + fr;
                    if (k < K) slab[(int64_t)n * K + k] = acc[i][j][r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small glue kernels of the training graph (HBM-bound, 16 bytes per thread)
// ---------------------------------------------------------------------------------------------------------------------
// out = a x + b y (y may be NULL)
__global__ __launch_bounds__(256) void axpby_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Y, float a, float b,
                                                     int64_t n8, uint16_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8];
        unpack8(*(const uint4*)(X + i * 8), x);
        if (Y) {
            unpack8(*(const uint4*)(Y + i * 8), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = a * x[e] + b * y[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= a;
        }
        *(uint4*)(out + i * 8) = pack8(x);
    }
}

// SiLU forward / backward on 16-bit tensors: mode 0: out = silu(x);  mode 1: out = dy * silu'(x)
__global__ __launch_bounds__(256) void silu_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ dY, int mode,
                                                    int64_t n8, uint16_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8], d[8];
        unpack8(*(const uint4*)(X + i * 8), x);
        if (mode) unpack8(*(const uint4*)(dY + i * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = mode ? gnb_dz(d[e], x[e], 1) : silu_f(x[e]);
        *(uint4*)(out + i * 8) = pack8(x);
    }
}

// partial[block] = sum over the block's elements of x * (y - z)   (z may be NULL); fixed order
__global__ __launch_bounds__(256) void dot_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Y,
                                                   const uint16_t* __restrict__ Z, int64_t n8, float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float x[8], y[8], z[8];
        unpack8(*(const uint4*)(X + i * 8), x);
        unpack8(*(const uint4*)(Y + i * 8), y);
        if (Z) unpack8(*(const uint4*)(Z + i * 8), z);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = __builtin_fmaf(x[e], Z ? y[e] - z[e] : y[e], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// Structured row maps on token-major [rows][C] tensors (the training graph's re-orderings; rows that map to nothing are zero):
//   0 UPSAMPLE2X  out (n, 2H, 2W) <- in (n, H, W), nearest             (Upsample2D before its conv, upsampling.py:142-186)
//   1 DILATE2X    out (n, 2H, 2W): in (n, H, W) at the even positions   (gradient of the stride-2 conv as a stride-1 conv)
//   2 SUMPOOL2X2  out (n, H, W)   <- sum of the 2x2 blocks of in (n, 2H, 2W)   (gradient of the nearest upsample)
//   3 T_TO_SEQ    out ((b, s), Tp, C) <- in ((b, t), s, C), zero rows for t >= T   (temporal attention: one sequence per site)
//   4 SEQ_TO_T    the inverse (pad rows dropped)
__global__ __launch_bounds__(256) void row_map_kernel(const uint16_t* __restrict__ X, int mode, int64_t out_rows, int C, int H, int W,
                                                       int T, int Tp, int S, uint16_t* __restrict__ out) {
    const int chunks = C >> 3;
    const int64_t total = out_rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int ch = (int)(i - r * chunks);
        int64_t src = -1;
        if (mode == 0 || mode == 1) {
            const int W2 = 2 * W, HW2 = 4 * H * W;
            const int64_t n = r / HW2;
            const int rem = (int)(r - n * HW2), y = rem / W2, x = rem - y * W2;
            if (mode == 0 || ((y & 1) == 0 && (x & 1) == 0)) src = n * (H * W) + (int64_t)(y >> 1) * W + (x >> 1);
        } else if (mode == 2) {
            const int64_t n = r / (H * W);
            const int rem = (int)(r - n * (H * W)), y = rem / W, x = rem - y * W;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float v[8];
                unpack8(*(const uint4*)(X + (n * (4 * H * W) + (int64_t)(2 * y + (a >> 1)) * (2 * W) + 2 * x + (a & 1)) * C + ch * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
            *(uint4*)(out + r * C + ch * 8) = pack8(acc);
            continue;
        } else if (mode == 3) {
            const int t = (int)(r % Tp);
            const int64_t bs = r / Tp, b = bs / S, sidx = bs - b * S;
            if (t < T) src = (b * T + t) * S + sidx;
        } else {
            const int64_t bt = r / S, sidx = r - bt * S, b = bt / T, t = bt - b * T;
            src = (b * S + sidx) * Tp + t;
        }
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (src >= 0) v = *(const uint4*)(X + src * C + ch * 8);
        *(uint4*)(out + r * C + ch * 8) = v;
    }
}

}  // namespace

static inline unsigned wiw_ew_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b > 256 * 32 ? 256 * 32 : (b < 1 ? 1 : b));
}

extern "C" int wiw_axpby_bf16(void* stream, const void* X, const void* Y, float a, float b, int64_t n, void* out) {
    WIW_REQUIRE(X && out && n > 0 && n % 8 == 0, "axpby: n must be a positive multiple of 8");
    hipLaunchKernelGGL(axpby_kernel, dim3(wiw_ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       (const uint16_t*)Y, a, b, n / 8, (uint16_t*)out);
    return wiw_check_launch("wiw_axpby_bf16");
}

extern "C" int wiw_silu_bf16(void* stream, const void* X, const void* dY, int backward, int64_t n, void* out) {
    WIW_REQUIRE(X && out && n > 0 && n % 8 == 0 && (!backward || dY), "silu: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(wiw_ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       (const uint16_t*)dY, backward, n / 8, (uint16_t*)out);
    return wiw_check_launch("wiw_silu_bf16");
}

extern "C" int wiw_dot_bf16(void* stream, const void* X, const void* Y, const void* Z, int64_t n, float* partial, int n_partial) {
    WIW_REQUIRE(X && Y && partial && n > 0 && n % 8 == 0 && n_partial > 0 && n_partial <= 4096, "dot: bad arguments");
    hipLaunchKernelGGL(dot_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X, (const uint16_t*)Y,
                       (const uint16_t*)Z, n / 8, partial);
    return wiw_check_launch("wiw_dot_bf16");
}

extern "C" int wiw_row_map_bf16(void* stream, const void* X, int mode, int64_t out_rows, int C, int H, int Wd, int T, int Tp, int S,
                                void* out) {
    WIW_REQUIRE(X && out && out_rows > 0 && C > 0 && C % 8 == 0 && mode >= 0 && mode <= 4, "row_map: bad arguments");
    hipLaunchKernelGGL(row_map_kernel, dim3(wiw_ew_blocks(out_rows * (C >> 3))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       mode, out_rows, C, H, Wd, T, Tp, S, (uint16_t*)out);
    return wiw_check_launch("wiw_row_map_bf16");
}

namespace {
}  // namespace

static int attn_bwd_launch(void* stream, const void* QKV, int ld, int k_off, int v_off, const void* Qt, const void* Kt,
                           const void* dOt, int64_t ldt, const void* O, const void* dO, int ldo, void* dQKV, int ldd,
                           float* lse, float* dsum, int seqs, int S, int Sp, int heads, int head_dim, float scale, int lse_given) {
    const bool tiled = S == Sp && S % 32 == 0 && S >= 128;   // the LDS-tiled kernels read Q, K, dO row-major only
    WIW_REQUIRE(!lse_given || tiled, "attn_bwd: a given log-sum-exp is taken by the LDS-tiled kernels only (S == Sp, S % 32 == 0, S >= 128)");
    WIW_REQUIRE(QKV && O && dO && dQKV && lse && dsum && (tiled || (Qt && Kt && dOt)), "attn_bwd: null pointer");
    WIW_REQUIRE(seqs > 0 && S > 0 && heads > 0 && Sp >= S && Sp % 16 == 0, "attn_bwd: bad sizes (Sp % 16 == 0)");
    WIW_REQUIRE(head_dim == 64, "attn_bwd: head_dim 64 only");
    WIW_REQUIRE(ld % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && (tiled || ldt % 4 == 0) && ldo % 8 == 0 && ldd % 4 == 0,
                "attn_bwd: misaligned strides");
    const int tiles = Sp / 16;
    const int64_t total = (int64_t)seqs * heads * tiles;
    WIW_REQUIRE(total < (1ll << 32), "attn_bwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const float LOG2E_ = 1.4426950408889634f;
    if (tiled) {                                       // long (spatial) sequences: the LDS-tiled kernels, 128 rows per workgroup
        const int blocks128 = (S + 127) / 128;
        const int64_t grid = (int64_t)seqs * heads * blocks128;
        hipLaunchKernelGGL((attn_bwd_dq_tiled_kernel<64>), dim3((unsigned)grid), dim3(256), 0, s, (const uint16_t*)QKV, ld, k_off,
                           v_off, (const uint16_t*)O, (const uint16_t*)dO, ldo, (uint16_t*)dQKV, ldd, lse, dsum, S, heads, blocks128,
                           scale, scale * LOG2E_, lse_given);
        hipLaunchKernelGGL((attn_bwd_dkv_tiled_kernel<64>), dim3((unsigned)grid), dim3(256), 0, s, (const uint16_t*)QKV, ld, k_off,
                           v_off, (const uint16_t*)dO, ldo, (uint16_t*)dQKV, ldd, lse, dsum, S, heads, blocks128, scale,
                           scale * LOG2E_);
        return wiw_check_launch("wiw_attn_bwd_bf16");
    }
    hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, (const uint16_t*)QKV, ld, k_off,
                       v_off, (const uint16_t*)Kt, ldt, (const uint16_t*)O, (const uint16_t*)dO, ldo, (uint16_t*)dQKV, ldd, lse, dsum,
                       S, Sp, heads, tiles, total, scale, scale * LOG2E_);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<64>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, (const uint16_t*)QKV, ld, k_off,
                       v_off, (const uint16_t*)Qt, (const uint16_t*)dOt, ldt, (const uint16_t*)dO, ldo, (uint16_t*)dQKV, ldd, lse,
                       dsum, S, Sp, heads, tiles, total, scale, scale * LOG2E_);
    return wiw_check_launch("wiw_attn_bwd_bf16");
}

extern "C" int wiw_attn_bwd_bf16(void* stream, const void* QKV, int ld, int k_off, int v_off, const void* Qt, const void* Kt,
                                 const void* dOt, int64_t ldt, const void* O, const void* dO, int ldo, void* dQKV, int ldd,
                                 float* lse, float* dsum, int seqs, int S, int Sp, int heads, int head_dim, float scale) {
    return attn_bwd_launch(stream, QKV, ld, k_off, v_off, Qt, Kt, dOt, ldt, O, dO, ldo, dQKV, ldd, lse, dsum, seqs, S, Sp, heads,
                           head_dim, scale, 0);
}

extern "C" int wiw_attn_bwd_given_lse_bf16(void* stream, const void* QKV, int ld, int k_off, int v_off, const void* O, const void* dO,
                                           int ldo, void* dQKV, int ldd, const float* lse, float* dsum, int seqs, int S, int heads,
                                           int head_dim, float scale) {
    return attn_bwd_launch(stream, QKV, ld, k_off, v_off, nullptr, nullptr, nullptr, 0, O, dO, ldo, dQKV, ldd, (float*)lse, dsum, seqs,
                           S, S, heads, head_dim, scale, 1);
}

extern "C" int wiw_geglu_fwd(void* stream, const void* P, int64_t rows, int Ch, void* H) {
    WIW_REQUIRE(P && H && rows > 0 && Ch > 0 && Ch % 8 == 0, "geglu_fwd: bad arguments");
    int64_t blocks = (rows * (Ch >> 3) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)P, rows, Ch,
                       (uint16_t*)H);
    return wiw_check_launch("wiw_geglu_fwd");
}

extern "C" int wiw_gather_taps_bf16(void* stream, const void* X, int64_t M, int C, int H, int Wd, int T, int temporal, int stride,
                                    void* Xcol) {
    WIW_REQUIRE(X && Xcol && M > 0 && C > 0 && C % 8 == 0 && H > 0 && Wd > 0 && (stride == 1 || (stride == 2 && !temporal)),
                "gather_taps: bad arguments");
    WIW_REQUIRE(M % ((int64_t)H * Wd) == 0 && (!temporal || (T > 0 && (M / ((int64_t)H * Wd)) % T == 0)), "gather_taps: bad geometry");
    int64_t blocks = (M * (temporal ? 3 : 9) * (C >> 3) + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(gather_taps_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X, M, C, H,
                       Wd, T, temporal, stride, (uint16_t*)Xcol);
    return wiw_check_launch("wiw_gather_taps_bf16");
}

extern "C" int wiw_gather_taps_t_bf16(void* stream, const void* X, int64_t M, int64_t Mp, int C, int H, int Wd, int T, int temporal,
                                      int stride, void* XcolT) {
    WIW_REQUIRE(X && XcolT && M > 0 && Mp >= M && Mp % 8 == 0 && C > 0 && C % 8 == 0 && H > 0 && Wd > 0 &&
                (stride == 1 || (stride == 2 && !temporal)), "gather_taps_t: bad arguments");
    WIW_REQUIRE(M % ((int64_t)H * Wd) == 0 && (!temporal || (T > 0 && (M / ((int64_t)H * Wd)) % T == 0)), "gather_taps_t: bad geometry");
    const int taps = temporal ? 3 : 9;
    const int64_t mblocks = (Mp + 63) / 64;
    WIW_REQUIRE(mblocks < (1ll << 31) && (int64_t)taps * ((C + 63) / 64) <= 65535, "gather_taps_t: grid too large");
    hipLaunchKernelGGL(gather_taps_t_kernel, dim3((unsigned)mblocks, (unsigned)(taps * ((C + 63) / 64))), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)X, M, Mp, C, H, Wd, T, temporal, stride, (uint16_t*)XcolT);
    return wiw_check_launch("wiw_gather_taps_t_bf16");
}

extern "C" int wiw_groupnorm_bwd(void* stream, const void* X, const void* dY, const float* stats, const float* gamma,
                                 const float* beta, int64_t rows, int C, int rows_per_unit, float eps, int silu, void* dX,
                                 float* unit_cs, float* AB, float* partial, int rows_per_block) {
    WIW_REQUIRE(X && dY && stats && gamma && beta && dX && unit_cs && AB && partial, "groupnorm_bwd: null pointer");
    WIW_REQUIRE(C > 0 && C % 8 == 0 && C % GNB_GROUPS == 0 && C <= 4096, "groupnorm_bwd: C must be %8, %32 and <= 4096");
    WIW_REQUIRE(rows > 0 && rows_per_unit > 0 && rows % rows_per_unit == 0 && rows_per_block > 0, "groupnorm_bwd: bad rows");
    const int units = (int)(rows / rows_per_unit);
    const int splits = (rows_per_unit + rows_per_block - 1) / rows_per_block;
    WIW_REQUIRE(splits <= 65535 && units <= 65535, "groupnorm_bwd: too many row splits / units");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gnb_reduce_kernel, dim3(splits, units), dim3(256), 0, s, (const uint16_t*)X, (const uint16_t*)dY, stats,
                       gamma, beta, C, rows_per_unit, rows_per_block, eps, silu, partial);
    hipLaunchKernelGGL(gnb_finish_kernel, dim3(units, GNB_GROUPS), dim3(256), 0, s, partial, splits, C, gamma, unit_cs, AB);
    hipLaunchKernelGGL(gnb_apply_kernel, dim3(splits, units), dim3(256), 0, s, (const uint16_t*)X, (const uint16_t*)dY, stats,
                       AB, gamma, beta, C, rows_per_unit, rows_per_block, eps, silu, (uint16_t*)dX);
    return wiw_check_launch("wiw_groupnorm_bwd");
}

extern "C" int wiw_wgrad_tn_bf16(void* stream, const void* dY, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K,
                                 int splits, float* slabs) {
    WIW_REQUIRE(dY && X && slabs && M > 0 && N > 0 && K > 0 && splits > 0 && splits <= 65535, "wgrad_tn: bad arguments");
    WIW_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K, "wgrad_tn: N, K, ldy, ldx % 8");
    WIW_REQUIRE(((((uintptr_t)dY) | ((uintptr_t)X)) & 15) == 0, "wgrad_tn: operands must be 16-byte aligned");
    int64_t rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;                                  // whole 32-row steps per split
    WIW_REQUIRE((int64_t)(splits - 1) * rps < M, "wgrad_tn: more splits than 32-row steps");
    const int64_t tiles = (int64_t)((N + WTN_BN - 1) / WTN_BN) * ((K + WTN_BK - 1) / WTN_BK);
    WIW_REQUIRE(tiles < (1ll << 31), "wgrad_tn: grid too large");
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dY,
                       ldy, (const uint16_t*)X, ldx, M, N, K, rps, slabs);
    return wiw_check_launch("wiw_wgrad_tn_bf16");
}

extern "C" int wiw_colsum(void* stream, const void* X, int is_f32, int64_t rows, int C, int parts, float* out) {
    WIW_REQUIRE(X && out && rows > 0 && C > 0 && parts > 0 && parts <= 65535, "colsum: bad arguments");
    WIW_REQUIRE(C % (is_f32 ? 4 : 8) == 0 && (((uintptr_t)X) & 15) == 0, "colsum: C must be a multiple of 8 (fp32: 4), X 16-byte aligned");
    const int64_t rpp = (rows + parts - 1) / parts;
    const int cw = is_f32 ? 4 : 8;
    const dim3 grid((unsigned)((C + 32 * cw - 1) / (32 * cw)), (unsigned)parts);
    if (is_f32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)X, rows, C, rpp, out);
    else hipLaunchKernelGGL(colsum_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X, rows, C, rpp, out);
    return wiw_check_launch("wiw_colsum");
}

extern "C" int64_t wiw_layernorm_bwd_partials(int64_t rows) {   // number of [2][C] fp32 partial slabs the kernel writes
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    return blocks * 4;
}

extern "C" int wiw_layernorm_bwd(void* stream, const void* X, const void* dY, const float* gamma, int64_t rows, int C, float eps,
                                 const void* dRes, void* dX, float* partial) {
    WIW_REQUIRE(X && dY && gamma && dX && partial, "layernorm_bwd: null pointer");
    WIW_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= LNB_MAXCH * 512, "layernorm_bwd: C must be %8 and <= 2048");
    const int64_t blocks = wiw_layernorm_bwd_partials(rows) / 4;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       (const uint16_t*)dY, gamma, rows, C, eps, (const uint16_t*)dRes, (uint16_t*)dX, partial);
    return wiw_check_launch("wiw_layernorm_bwd");
}

extern "C" int wiw_geglu_bwd(void* stream, const void* P, const void* dH, int64_t rows, int Ch, void* dP) {
    WIW_REQUIRE(P && dH && dP && rows > 0 && Ch > 0 && Ch % 8 == 0, "geglu_bwd: bad arguments");
    int64_t blocks = (rows * (Ch >> 3) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)P,
                       (const uint16_t*)dH, rows, Ch, (uint16_t*)dP);
    return wiw_check_launch("wiw_geglu_bwd");
}

// EMAModel.step on a flat range: s -= omd * (s - p) with the three roundings of the reference's tensor expression (no fma
// contraction), 16 bytes per lane, grid-stride
WIW_DEV float ema_one(float s, float p, float omd) {
    float d = omd * (s - p);
    asm volatile("" : "+v"(d));   // the product is ROUNDED before the subtraction: no fma contraction across this point
    return s - d;
}
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ s, const float* __restrict__ p, int64_t n, float omd) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 sv = ((const float4*)s)[i];
        const float4 pv = ((const float4*)p)[i];
        sv.x = ema_one(sv.x, pv.x, omd); sv.y = ema_one(sv.y, pv.y, omd);
        sv.z = ema_one(sv.z, pv.z, omd); sv.w = ema_one(sv.w, pv.w, omd);
        ((float4*)s)[i] = sv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        s[i] = ema_one(s[i], p[i], omd);
    }
}

extern "C" int wiw_ema_step_f32(void* stream, float* shadow, const float* param, int64_t n, float one_minus_decay) {
    WIW_REQUIRE(shadow && param && n > 0, "ema_step: null pointer or n <= 0");
    WIW_REQUIRE((((uintptr_t)shadow | (uintptr_t)param) & 15) == 0, "ema_step: pointers must be 16-byte aligned");
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, shadow, param, n, one_minus_decay);
    return wiw_check_launch("wiw_ema_step_f32");
}

extern "C" int wiw_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step) {
    WIW_REQUIRE(p && g && m && v, "adamw: null pointer");
    WIW_REQUIRE(n > 0 && step >= 1, "adamw: n > 0 and step >= 1");
    WIW_REQUIRE(((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && ((((uintptr_t)p16) & 7) == 0),
                "adamw: pointers must be 16-byte aligned (8 for the 16-bit copy)");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (uint16_t*)p16, n, lr,
                       beta1, beta2, eps, weight_decay, 1.0f / bc1, 1.0f / sqrtf(bc2));
    return wiw_check_launch("wiw_adamw_step");
}

extern "C" int wiw_edm_loss_grad(void* stream, const float* pred, const float* noisy, const float* target, int64_t n,
                                 float sigma, float* grad, float* partial, int n_partial) {
    WIW_REQUIRE(pred && noisy && target && grad && partial, "edm_loss: null pointer");
    WIW_REQUIRE(n > 0 && sigma > 0.0f && n_partial > 0 && n_partial <= 4096, "edm_loss: bad sizes");
    const float s2 = sigma * sigma;
    hipLaunchKernelGGL(edm_loss_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, pred, noisy, target, n,
                       -sigma / sqrtf(s2 + 1.0f), 1.0f / (s2 + 1.0f), (1.0f + s2) / s2, grad, partial);
    return wiw_check_launch("wiw_edm_loss_grad");
}
