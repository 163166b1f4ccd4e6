// Library bookkeeping + the small HBM-bound kernels around the UNet: conditioning-embedding combine,
// UNet input assembly (scale_model_input + CFG doubling + channel concat, NCHW fp32 -> NHWC bf16)
// and the fused CFG-combine + Euler step.  All are pure streaming kernels: coalesced 16-byte
// accesses on the wide side, grid-stride, no LDS.
#include <stdio.h>
#include <string.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// error bookkeeping
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";

void wiw_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int wiw_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return WIW_ELAUNCH;
    }
    return WIW_OK;
}

extern "C" int wiw_abi_version(void) { return WIW_ABI_VERSION; }
extern "C" int wiw_dtype(void) { return WIW_DTYPE_CODE; }
extern "C" const char* wiw_last_error(void) { return g_err; }

extern "C" int wiw_device_check(int dev, char* name, int name_len) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        wiw_set_error("no HIP device");
        return WIW_ENODEV;
    }
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return WIW_ENODEV;
    }
    return WIW_OK;
}

// ---------------------------------------------------------------------------------------------
namespace {

__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// out[(i*T + t)][c] = silu(time[i][c] + act[((i % B)*T + t)][c] + noise[i][c])      (bf16)
__global__ void emb_combine_kernel(const float* __restrict__ time, const float* __restrict__ act,
                                   const float* __restrict__ noise, int Bc, int B, int T, int E, uint16_t* out) {
    const int64_t total = (int64_t)Bc * T * E;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % E);
        const int row = (int)(idx / E);
        const int i = row / T, t = row - i * T;
        const float v = time[(int64_t)i * E + c] + act[((int64_t)(i % B) * T + t) * E + c] + noise[(int64_t)i * E + c];
        out[idx] = f2bf(silu_f(v));
    }
}

// One thread per output pixel row (NHWC, Cpad channels = Cpad/8 chunks of 16 B).
__global__ void prep_input_kernel(const float* __restrict__ lat, const float* __restrict__ img, int B, int T, int hw,
                                  float inv_scale, int Cpad, uint16_t* X) {
    const int chunks = Cpad >> 3;
    const int64_t total = (int64_t)2 * B * T * hw * chunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const int64_t row = idx / chunks;          // (i*T + t)*hw + p
        uint4 o = uint4{0u, 0u, 0u, 0u};
        if (ch == 0) {
            const int p = (int)(row % hw);
            const int64_t ft = row / hw;           // i*T + t
            const int t = (int)(ft % T), i = (int)(ft / T);
            const int b = i % B;
            float v[8];
            const float* l = lat + (((int64_t)b * T + t) * 4) * hw + p;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = l[(int64_t)c * hw] * inv_scale;
            if (i >= B) {
                const float* im = img + ((int64_t)b * 4) * hw + p;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 + c] = im[(int64_t)c * hw];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[4 + c] = 0.f;
            }
            o = pack8(v);
        }
        *(uint4*)(X + row * Cpad + ch * 8) = o;
    }
}

// One thread per (b, t, p): 4 channels.
__global__ void cfg_euler_kernel(const float* __restrict__ V, int ldv, float* lat, int B, int T, int hw, float sigma,
                                 float sigma_next, float gmin, float gmax) {
    const int64_t total = (int64_t)B * T * hw;
    const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
    const float c_skip = 1.0f / (sigma * sigma + 1.0f);
    const float dt = sigma_next - sigma;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx % hw);
        const int64_t bt = idx / hw;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const float g = (T > 1) ? gmin + (gmax - gmin) * (float)t / (float)(T - 1) : gmin;
        const float* vu = V + (((int64_t)b * T + t) * hw + p) * ldv;
        const float* vc = V + ((((int64_t)(B + b)) * T + t) * hw + p) * ldv;
        float* x = lat + (((int64_t)b * T + t) * 4) * hw + p;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = vu[c] + g * (vc[c] - vu[c]);
            const float xv = x[(int64_t)c * hw];
            const float x0 = v * c_out + xv * c_skip;
            x[(int64_t)c * hw] = xv + (xv - x0) / sigma * dt;
        }
    }
}

inline int grid_for(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    if (g > 2048 * 8) g = 2048 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// Y[c][r] = X[r][c0 + c]  (16-bit), 64 x 64 tiles through LDS: 128-byte row segments on both sides.
// Hands the attention kernel V^T from a V that one fused q|k|v projection wrote row-major, and the training step the
// K-contiguous operands of its weight-gradient GEMMs.  A thread loads the same 8 columns of TWO consecutive rows and
// writes 8 words (X[2p][c] | X[2p+1][c] << 16) into tile[c][p] — the 2 x 2 transposes happen in registers, so LDS sees
// 32-bit writes (conflict-free at a row stride of 33 words) and 32-bit reads instead of twice as many 16-bit ones.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ X, int64_t ldx, int c0, int64_t rows,
                                                              int C, uint16_t* __restrict__ Y, int64_t ldy) {
    __shared__ uint32_t tile[64][33];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int cb = blockIdx.y * 64;
    const int tid = threadIdx.x;
    {
        const int p = tid >> 3, ch = tid & 7;                     // row pair, 16-byte chunk
        uint4 a = uint4{0u, 0u, 0u, 0u}, b = a;
        if (cb + ch * 8 < C && r0 + 2 * p < rows) {               // rows % 8 == 0: the pair is inside or outside together
            const uint16_t* src = X + (r0 + 2 * p) * ldx + c0 + cb + ch * 8;
            a = *(const uint4*)src;
            b = *(const uint4*)(src + ldx);
        }
        const uint32_t au[4] = {a.x, a.y, a.z, a.w}, bu[4] = {b.x, b.y, b.z, b.w};
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
        const int c = i >> 3, ch = i & 7;                         // output row c, 8 consecutive source rows = 4 words
        if (cb + c < C && r0 + ch * 8 < rows)
            *(uint4*)(Y + (int64_t)(cb + c) * ldy + r0 + ch * 8) =
                uint4{tile[c][ch * 4], tile[c][ch * 4 + 1], tile[c][ch * 4 + 2], tile[c][ch * 4 + 3]};
    }
}

}  // namespace

extern "C" int wiw_transpose_bf16(void* stream, const void* X, int64_t ldx, int c0, int64_t rows, int C, void* Y, int64_t ldy) {
    WIW_REQUIRE(X && Y, "transpose: null pointer");
    WIW_REQUIRE(rows > 0 && C > 0 && rows % 8 == 0 && C % 8 == 0 && c0 % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0,
                "transpose: rows, C, c0, ldx, ldy must be multiples of 8");
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((C + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)X, ldx, c0, rows, C, (uint16_t*)Y, ldy);
    return wiw_check_launch("wiw_transpose_bf16");
}

namespace {
// out[i] = round16(X[i]): the 16-bit MFMA operand of an fp32 residual-stream tensor (ABI 11); 8 values per thread
__global__ __launch_bounds__(256) void cast_f32_to_16_kernel(const float* __restrict__ X, int64_t n8, uint16_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = *(const float4*)(X + i * 8), b = *(const float4*)(X + i * 8 + 4);
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        *(uint4*)(out + i * 8) = pack8(f);
    }
}
}  // namespace

extern "C" int wiw_cast_f32_to_16(void* stream, const float* X, int64_t n, void* out) {
    WIW_REQUIRE(X && out && n > 0 && n % 8 == 0, "cast_f32_to_16: n must be a positive multiple of 8");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(cast_f32_to_16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, n / 8, (uint16_t*)out);
    return wiw_check_launch("wiw_cast_f32_to_16");
}

namespace {
// Box calibration (bench.py `box.mfma_peak_measured`): nothing but back-to-back MFMAs on register operands — 8 independent
// accumulators per wave, 8 waves per CU (two per SIMD), one block per CU.  What THIS box's clocks give the matrix pipe; the
// boxes of the pool differ by +-5 % on every kernel, and a reader can normalise a bench line by this number.
__global__ __launch_bounds__(512, 2) void calib_mfma_kernel(int iters, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (short)(WIW_ONE16 + ((lane + e) & 3)); b[e] = (short)(WIW_ONE16 - ((lane * 3 + e) & 3)); }
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = WIW_MFMA(a, b, acc[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) out[0] = s;   // never true: keeps the loop alive
}
}  // namespace

namespace {
// ... and the same loop on RANDOM operands (src: 128 x 16 bytes of the library's 16-bit type, N(0, 0.5): lane l takes A from
// entry l, B from entry 64 + l).  Power follows the data: held for a second this is what the board's power management lets
// the matrix pipe SUSTAIN (bench.py `box.mfma_sustained`), where the near-constant operands of the burst above toggle little.
__global__ __launch_bounds__(512, 2) void calib_mfma_random_kernel(const bf16x8* __restrict__ src, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const bf16x8 a = src[lane], b = src[64 + lane];
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = WIW_MFMA(a, b, acc[j]);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678f) out[0] = s;   // never true: keeps the loop alive
}
}  // namespace

extern "C" int wiw_calib_mfma_random(void* stream, int blocks, int iters, const void* src, float* out) {
    WIW_REQUIRE(blocks > 0 && iters > 0 && src != nullptr && out != nullptr && (((uintptr_t)src) & 15) == 0, "calib_mfma_random: bad arguments");
    hipLaunchKernelGGL(calib_mfma_random_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, (const bf16x8*)src, iters, out);
    return wiw_check_launch("wiw_calib_mfma_random");
}

/* Box calibration: `blocks` blocks of 8 waves run `iters` x 8 MFMAs (16x16x32, 16 384 flop each) per wave; the caller times
 * it with events: flops = blocks * 8 * iters * 8 * 16384. */
extern "C" int wiw_calib_mfma(void* stream, int blocks, int iters, float* out) {
    WIW_REQUIRE(blocks > 0 && iters > 0 && out != nullptr, "calib_mfma: bad arguments");
    hipLaunchKernelGGL(calib_mfma_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, iters, out);
    return wiw_check_launch("wiw_calib_mfma");
}

extern "C" int wiw_fill_f32(void* stream, float* p, int64_t n, float value) {
    WIW_REQUIRE(p != nullptr && n > 0, "fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, n, value);
    return wiw_check_launch("wiw_fill_f32");
}

extern "C" int wiw_emb_combine(void* stream, const float* time, const float* act, const float* noise, int Bc, int B,
                               int T, int E, void* out) {
    WIW_REQUIRE(time && act && noise && out, "emb_combine: null pointer");
    WIW_REQUIRE(Bc > 0 && B > 0 && Bc % B == 0 && T > 0 && E > 0, "emb_combine: bad sizes");
    hipLaunchKernelGGL(emb_combine_kernel, dim3(grid_for((int64_t)Bc * T * E, 256)), dim3(256), 0, (hipStream_t)stream,
                       time, act, noise, Bc, B, T, E, (uint16_t*)out);
    return wiw_check_launch("wiw_emb_combine");
}

extern "C" int wiw_prep_unet_input(void* stream, const float* latents, const float* image_latents, int B, int T,
                                   int hw, float sigma, int Cpad, void* X) {
    WIW_REQUIRE(latents && image_latents && X, "prep_unet_input: null pointer");
    WIW_REQUIRE(B > 0 && T > 0 && hw > 0 && Cpad >= 8 && Cpad % 8 == 0, "prep_unet_input: bad sizes");
    const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
    const int64_t total = (int64_t)2 * B * T * hw * (Cpad / 8);
    hipLaunchKernelGGL(prep_input_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, latents,
                       image_latents, B, T, hw, inv, Cpad, (uint16_t*)X);
    return wiw_check_launch("wiw_prep_unet_input");
}

extern "C" int wiw_cfg_euler_step(void* stream, const float* V, int ldv, float* latents, int B, int T, int hw,
                                  float sigma, float sigma_next, float gmin, float gmax) {
    WIW_REQUIRE(V && latents, "cfg_euler_step: null pointer");
    WIW_REQUIRE(B > 0 && T > 0 && hw > 0 && ldv >= 4 && sigma > 0.f, "cfg_euler_step: bad sizes");
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid_for((int64_t)B * T * hw, 256)), dim3(256), 0, (hipStream_t)stream, V,
                       ldv, latents, B, T, hw, sigma, sigma_next, gmin, gmax);
    return wiw_check_launch("wiw_cfg_euler_step");
}
