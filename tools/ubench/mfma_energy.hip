// Round 6: energy per TFLOP of the two 16-bit MFMA shapes under the 256x320 GEMM kernel's OWN operand
// pattern.  gemm_huge.hip's wave tile is 64 rows x 160 columns (160 accumulator VGPRs, 8 waves per CU = 2 per SIMD); per
// 32-deep k-step a wave reads 4 A + 10 W fragments (14 ds_read_b128) and issues
//     S16: 40 x v_mfma_f32_16x16x32   (8 operand VGPRs + a 4-register accumulator per  8 192 MACs)
//     S32: 20 x v_mfma_f32_32x32x16   (8 operand VGPRs + a 16-register accumulator per 16 384 MACs; 2 x (2 A + 5 W) fragments)
// The LDS bytes per FLOP are the same; what differs is how often the register file is read per MAC.  The kernel is held for
// seconds (one launch = `iters` k-steps, launches back to back) on random N(0,1) operands (power follows the data: zeros cost
// little) while tools/mfma_energy.py samples rocm-smi; it prints the sustained TFLOP/s of the whole run.
//   variants: 0 = S16 + LDS reads, 1 = S32 + LDS reads, 2 = S16 registers only, 3 = S32 registers only,
//             4 / 5 = S16 / S32 + LDS reads with fp16 operands
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_energy.hip -o /tmp/mfma_energy && /tmp/mfma_energy <variant> <seconds>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__

template <bool F16> DEV void mfma16(f4v& c, s8 a, s8 b) {
    if (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool F16> DEV void mfma32(f16v& c, s8 a, s8 b) {
    if (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(off))

constexpr int SMEM = 147456;   // 144 KiB: one block per CU, as the GEMM (its two 72-KiB ring stages)

template <int SHAPE, bool READS, bool F16>
__global__ __launch_bounds__(512, 2) void k(const uint4* src, int iters, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // LDS <- random operands (every block the same 144 KiB; the source is L2-resident after the first block)
    for (int i = threadIdx.x; i < SMEM / 16; i += 512) ((uint4*)smem)[i] = src[i];
    __syncthreads();
    // conflict-free fragment image: row = lane & 15 (S16) / lane & 31 (S32), 16-byte chunk XOR-swizzled by the row; the
    // waves of a row group share A rows, those of a column group W rows, as in the GEMM (wm = wave & 3, wn = wave >> 2)
    const int wm = wave & 3, wn = wave >> 2;
    unsigned ra, rb;
    if (SHAPE == 16) {
        const int frow = lane & 15, fq = lane >> 4;
        ra = (unsigned)(uintptr_t)(smem + wm * 8192 + frow * 128 + ((fq ^ (frow & 7)) << 4));
        rb = (unsigned)(uintptr_t)(smem + 32768 + wn * 20480 + frow * 128 + ((fq ^ (frow & 7)) << 4));
    } else {
        const int frow = lane & 31, fq = lane >> 5;
        ra = (unsigned)(uintptr_t)(smem + wm * 8192 + frow * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));
        rb = (unsigned)(uintptr_t)(smem + 32768 + wn * 20480 + frow * 128 + ((fq ^ ((frow >> 1) & 7)) << 4));
    }
    float res = 0.f;
    if constexpr (SHAPE == 16) {
        f4v acc[4][10];
        for (int m = 0; m < 4; ++m)
            for (int j = 0; j < 10; ++j) acc[m][j] = f4v{0.f, 0.f, 0.f, 0.f};
        s8 pa[4], pb[4];
        for (int i = 0; i < 4; ++i) { pa[i] = *(const s8*)(smem + wave * 4096 + i * 1024 + lane * 16); pb[i] = *(const s8*)(smem + 65536 + wave * 4096 + i * 1024 + lane * 16); }
        for (int it = 0; it < iters; ++it) {
            const unsigned o = (unsigned)((it & 1) * 64);      // k-step 0 / 1 of the K tile: the other half of the 128-byte rows
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (READS) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(pb[j & 3]));
                mfma16<F16>(acc[0][j], pb[j & 3], pa[0]);
                mfma16<F16>(acc[1][j], pb[j & 3], pa[1]);
                mfma16<F16>(acc[2][j], pb[j & 3], pa[2]);
                mfma16<F16>(acc[3][j], pb[j & 3], pa[3]);
                if (READS) {
                    DSR(pb[j & 3], rb ^ o, ((j + 4) % 10) * 2048);      // 10 W fragments per k-step, three ahead of their MFMAs
                    if (j >= 6) DSR(pa[j - 6], ra ^ o, (j - 6) * 2048);  // 4 A fragments of the next k-step
                }
            }
        }
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int m = 0; m < 4; ++m)
            for (int j = 0; j < 10; ++j) res += acc[m][j][0] + acc[m][j][3];
        for (int i = 0; i < 4; ++i) res += (float)pa[i][0] + (float)pb[i][0];
    } else {
        f16v acc[2][5];
        for (int m = 0; m < 2; ++m)
            for (int j = 0; j < 5; ++j)
                for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
        s8 pa[2], pb[4];
        for (int i = 0; i < 2; ++i) pa[i] = *(const s8*)(smem + wave * 4096 + i * 1024 + lane * 16);
        for (int i = 0; i < 4; ++i) pb[i] = *(const s8*)(smem + 65536 + wave * 4096 + i * 1024 + lane * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {            // two 16-deep k-steps = the 32-deep step of the S16 form
                const unsigned o = (unsigned)(((it & 1) * 2 + half) * 32);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int w = (half * 5 + j) & 3;
                    if (READS) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(pb[w]));
                    mfma32<F16>(acc[0][j], pb[w], pa[0]);
                    mfma32<F16>(acc[1][j], pb[w], pa[1]);
                    if (READS) {
                        DSR(pb[w], rb ^ o, ((j + 4) % 5) * 4096);
                        if (j >= 3) DSR(pa[j - 3], ra ^ o, (j - 3) * 4096);
                    }
                }
            }
        }
        if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int m = 0; m < 2; ++m)
            for (int j = 0; j < 5; ++j) res += acc[m][j][0] + acc[m][j][15];
        for (int i = 0; i < 2; ++i) res += (float)pa[i][0];
        for (int i = 0; i < 4; ++i) res += (float)pb[i][0];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

static unsigned short to16(float f, bool f16) {
    if (f16) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
    unsigned u; __builtin_memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int SHAPE, bool READS, bool F16>
void run(const char* name, double seconds) {
    std::vector<unsigned short> h(SMEM / 2);
    unsigned s = 12345u;
    auto uni = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) + 0.5f) / 16777216.0f; };
    for (auto& v : h) v = to16(sqrtf(-2.f * logf(uni())) * cosf(6.2831853f * uni()) * 0.5f, F16);
    uint4* src; float* out;
    hipMalloc(&src, SMEM); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), SMEM, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<SHAPE, READS, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    const int iters = 40000;                                     // ~25-40 ms per launch
    hipLaunchKernelGGL((k<SHAPE, READS, F16>), dim3(256), dim3(512), SMEM, 0, src, iters, out);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((k<SHAPE, READS, F16>), dim3(256), dim3(512), SMEM, 0, src, iters, out);
        hipDeviceSynchronize();
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // per k-step and wave: 64 x 160 x 32 MACs
    const double tf = 2.0 * 64 * 160 * 32 * 8 * 256 * (double)iters * launches / el / 1e12;
    printf("%-40s %8.1f TFLOP/s sustained over %.1f s (%ld launches of %d k-steps)\n", name, tf, el, launches, iters);
    fflush(stdout);
    hipFree(src); hipFree(out);
}

int main(int argc, char** argv) {
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    const double sec = argc > 2 ? atof(argv[2]) : 4.0;
    switch (v) {
        case 0: run<16, true, false>("bf16 16x16x32 + 14 ds_read_b128 / 40 MFMA", sec); break;
        case 1: run<32, true, false>("bf16 32x32x16 + 14 ds_read_b128 / 20 MFMA", sec); break;
        case 2: run<16, false, false>("bf16 16x16x32 registers only", sec); break;
        case 3: run<32, false, false>("bf16 32x32x16 registers only", sec); break;
        case 4: run<16, true, true>("fp16 16x16x32 + 14 ds_read_b128 / 40 MFMA", sec); break;
        case 5: run<32, true, true>("fp16 32x32x16 + 14 ds_read_b128 / 20 MFMA", sec); break;
        default: return 1;
    }
    return 0;
}
