// Issue rate of VALU instruction kinds for ONE wave per SIMD (a 256-thread block per CU) and for two (512 threads), in cycles
// per wave-instruction, measured with s_memtime around 8 x 64 unrolled instructions (all CUs busy).  Question behind it
// (csrc/ffn.hip): what does the exact-erf GELU of 32 values per lane cost a lone wave, and which of its instructions to
// pack (v_pk_*) or to avoid (v_exp / v_rcp)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(float* out, long long* cyc, float seed) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const f2 c2 = f2{seed, seed};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 8; ++it) {
        if (KIND == 0) {          // 8 independent v_fma_f32 chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        } else if (KIND == 1) {   // one dependent v_fma_f32 chain
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[0]) : "v"(seed));)
        } else if (KIND == 2) {   // 8 independent v_pk_fma_f32 chains
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                              : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));)
        } else if (KIND == 3) {   // one dependent v_pk_fma_f32 chain
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[0]) : "v"(c2));)
        } else if (KIND == 4) {   // 8 independent v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                              "v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        } else if (KIND == 5) {   // 8 independent v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n"
                              "v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        } else if (KIND == 6) {   // v_exp_f32 with 3 independent v_fma_f32 after each (does the transcendental overlap plain VALU?)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n"
                              "v_exp_f32 %1, %1\n v_fma_f32 %7, %7, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(seed));)
        } else if (KIND == 7) {   // 8 independent v_cvt_pk_bf16_f32
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                              "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                              : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));)
        } else if (KIND == 8) {   // 8 independent v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                              : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, int threads, int per_it) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001f);
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d waves/SIMD: %6.2f cycles per wave-instruction\n", name, threads / 256, (double)h / (8.0 * per_it));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<0>("v_fma_f32, 8 independent chains", threads, 64);
        run<1>("v_fma_f32, dependent chain", threads, 64);
        run<2>("v_pk_fma_f32, 8 independent chains", threads, 64);
        run<3>("v_pk_fma_f32, dependent chain", threads, 64);
        run<8>("v_pk_mul_f32, 8 independent", threads, 64);
        run<4>("v_exp_f32, 8 independent", threads, 64);
        run<5>("v_rcp_f32, 8 independent", threads, 64);
        run<6>("v_exp_f32 + 3 v_fma_f32 (per 4 instr)", threads, 64);
        run<7>("v_cvt_pk_bf16_f32, 8 independent", threads, 64);
    }
    return 0;
}
