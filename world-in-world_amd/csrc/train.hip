// Fine-tuning step (FTsvd/train_svd.py:844-970) — FIRST kernels of that row: the optimiser update and
// the EDM loss with its gradient.  The backward kernels of the UNet operators are not built yet.
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

}  // namespace

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
