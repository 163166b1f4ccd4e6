// Kernels that exist only for the temporal VAE (the "next" rows f1 / f4):
//   * row softmax fp32 -> bf16 for the single-head, head_dim = C (512) attention of the VAE mid blocks, which is
//     evaluated as two MFMA GEMMs (scores = Q.K^T, out = P.V) around this kernel — a 512-wide head does not fit the
//     register budget of the flash kernel in attention.hip, and this attention is ~2 % of a decode;
//   * the decoder's last layer: (3,1,1) temporal convolution over 3 channels fused with the NHWC -> NCHW fp32
//     frame layout the worker hands to PIL.
// Both are HBM-streaming kernels: one read and one write per element, 16-byte accesses, fp32 math.
#include "common.h"

namespace {

constexpr int SM_THREADS = 256;
constexpr int SM_MAXV = 16;   // float4 per thread held in registers -> cols <= 256 * 4 * 16 = 16384

__global__ __launch_bounds__(SM_THREADS) void softmax_rows_kernel(const float* __restrict__ X, int64_t ldx, int cols,
                                                                   uint16_t* __restrict__ P, int64_t ldp) {
    __shared__ float red[2][SM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = X + (int64_t)blockIdx.x * ldx;
    uint16_t* p = P + (int64_t)blockIdx.x * ldp;
    const int n4 = cols >> 2;
    float4 v[SM_MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_MAXV; ++k) {
        const int i = tid + k * SM_THREADS;
        if (i < n4) {
            v[k] = *(const float4*)(x + 4 * i);
            mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[0][wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float L2E = 1.4426950408889634f;
    const float off = -mx * L2E;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAXV; ++k) {
        const int i = tid + k * SM_THREADS;
        if (i < n4) {
            v[k].x = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].x, L2E, off));
            v[k].y = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].y, L2E, off));
            v[k].z = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].z, L2E, off));
            v[k].w = __builtin_amdgcn_exp2f(__builtin_fmaf(v[k].w, L2E, off));
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[1][wave] = s;
    __syncthreads();
    const float inv = 1.0f / ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
#pragma unroll
    for (int k = 0; k < SM_MAXV; ++k) {
        const int i = tid + k * SM_THREADS;
        if (i < n4) {
            uint2 o;
            o.x = pack2bf(v[k].x * inv, v[k].y * inv);
            o.y = pack2bf(v[k].z * inv, v[k].w * inv);
            *(uint2*)(p + 4 * i) = o;
        }
    }
}

// Y fp32 [frames*HW][ldy] (channels 0..2) -> out fp32 [frames][3][HW];  frames = batch*T, zero padding over T
__global__ __launch_bounds__(256) void time_conv_out_kernel(const float* __restrict__ Y, int ldy,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             int frames, int T, int HW, float* __restrict__ out) {
    float wr[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) wr[i] = w[i];   // [co][ci][dt]
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    const int64_t total = (int64_t)frames * HW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / HW);
        const int pix = (int)(idx - (int64_t)n * HW);
        const int t = n % T;
        float a0 = b0, a1 = b1, a2 = b2;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int tt = t + dt - 1;
            if (tt < 0 || tt >= T) continue;
            const float4 y = *(const float4*)(Y + (idx + (int64_t)(dt - 1) * HW) * ldy);
            const float x0 = y.x, x1 = y.y, x2 = y.z;
            a0 += wr[0 * 9 + 0 * 3 + dt] * x0 + wr[0 * 9 + 1 * 3 + dt] * x1 + wr[0 * 9 + 2 * 3 + dt] * x2;
            a1 += wr[1 * 9 + 0 * 3 + dt] * x0 + wr[1 * 9 + 1 * 3 + dt] * x1 + wr[1 * 9 + 2 * 3 + dt] * x2;
            a2 += wr[2 * 9 + 0 * 3 + dt] * x0 + wr[2 * 9 + 1 * 3 + dt] * x1 + wr[2 * 9 + 2 * 3 + dt] * x2;
        }
        float* o = out + ((int64_t)n * 3) * HW + pix;
        o[0] = a0;
        o[HW] = a1;
        o[2 * (int64_t)HW] = a2;
    }
}

// NCHW fp32 (frames, Cin, HW) * scale -> NHWC bf16 [frames*HW][Cpad], channels >= Cin zero
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ X, int frames, int Cin, int HW,
                                                                float scale, int Cpad, uint16_t* __restrict__ out) {
    const int chunks = Cpad >> 3;
    const int64_t total = (int64_t)frames * HW * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const int64_t row = idx / chunks;
        const int n = (int)(row / HW);
        const int pix = (int)(row - (int64_t)n * HW);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = ch * 8 + e;
            f[e] = c < Cin ? X[((int64_t)n * Cin + c) * HW + pix] * scale : 0.f;
        }
        *(uint4*)(out + row * Cpad + ch * 8) = pack8(f);
    }
}

inline int grid_cap(int64_t total, int block, int cap) {
    int64_t g = (total + block - 1) / block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int wiw_softmax_rows_f32_bf16(void* stream, const float* X, int64_t ldx, int64_t rows, int cols, void* P,
                                         int64_t ldp) {
    WIW_REQUIRE(X && P, "softmax_rows: null pointer");
    WIW_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0, "softmax_rows: bad sizes");
    WIW_REQUIRE(cols % 4 == 0 && cols <= SM_THREADS * 4 * SM_MAXV, "softmax_rows: cols must be %4 and <= 16384");
    WIW_REQUIRE(ldx % 4 == 0 && ldp % 4 == 0 && ldx >= cols && ldp >= cols, "softmax_rows: strides must be %4 and >= cols");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(SM_THREADS), 0, (hipStream_t)stream, X, ldx, cols,
                       (uint16_t*)P, ldp);
    return wiw_check_launch("wiw_softmax_rows_f32_bf16");
}

extern "C" int wiw_vae_time_conv_out(void* stream, const float* Y, int ldy, const float* weight, const float* bias,
                                     int frames, int T, int HW, float* out) {
    WIW_REQUIRE(Y && weight && bias && out, "vae_time_conv_out: null pointer");
    WIW_REQUIRE(ldy >= 4 && ldy % 4 == 0 && frames > 0 && T > 0 && frames % T == 0 && HW > 0, "vae_time_conv_out: bad sizes");
    hipLaunchKernelGGL(time_conv_out_kernel, dim3(grid_cap((int64_t)frames * HW, 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, Y, ldy, weight, bias, frames, T, HW, out);
    return wiw_check_launch("wiw_vae_time_conv_out");
}

extern "C" int wiw_nchw_f32_to_nhwc_bf16(void* stream, const float* X, int frames, int Cin, int HW, float scale,
                                         int Cpad, void* out) {
    WIW_REQUIRE(X && out, "nchw_to_nhwc: null pointer");
    WIW_REQUIRE(frames > 0 && Cin > 0 && HW > 0 && Cpad >= Cin && Cpad % 8 == 0, "nchw_to_nhwc: bad sizes");
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(grid_cap((int64_t)frames * HW * (Cpad / 8), 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, X, frames, Cin, HW, scale, Cpad, (uint16_t*)out);
    return wiw_check_launch("wiw_nchw_f32_to_nhwc_bf16");
}
