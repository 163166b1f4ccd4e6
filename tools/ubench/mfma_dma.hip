// Round 5: what does an LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) cost a wave that is issuing MFMAs, and
// does it matter WHEN the four waves of a block (one per SIMD) issue theirs?  ffn32.hip's ablation priced one DMA piece at
// ~55 cycles of a wave's issue time (127 us on 296 us of MFMAs at 1 piece per 4 MFMAs) with every wave issuing in the same gap;
// the CU's fill path moves ~63 B/clk (tools/ubench/lds_fill.hip), i.e. 16 cycles per piece — if the 55 are three other waves'
// pieces queued in front, staggering the waves' DMA gaps should hide the cost in the 32-cycle shadow of a 32x32x16 MFMA.
//
// One block of 256 threads per CU (LDS sized so that a second block does not fit), every wave runs PERIODS x
//     { 20 MFMA 32x32x16  [+ 9 ds_read_b128 in the even gaps]  [+ ND DMA pieces] }          (one 16-k step of a 128x160 wave tile)
// DMA gaps: SAME = every wave in gaps 1, 5, 9, 13, (17);  STAG = wave w in gaps 1 + w + 4 i (w = 0..3), a wave-uniform branch
// in every gap in both forms;  SAME + barrier/4 = an s_barrier every 4 periods (a K tile of 64);  + skew = after that barrier
// wave w idles 16 w cycles, so that the four waves' pieces reach the fill path one piece time apart.  Source: P0 one contiguous KiB (tiled weights), P1 eight 128-byte row segments 2560 B apart (a
// row-major A operand); L2-resident 2 MiB shared by all blocks, or 64 MiB streamed (HBM / MALL).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_dma.hip -o /tmp/mdma && /tmp/mdma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define DEV __device__ __forceinline__

DEV void mfma32(f16v& c, s8 a, s8 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

template <bool READS, int ND, int STAG, int PAT>
__global__ __launch_bounds__(256) void k(const char* src, size_t region, int periods, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f16v acc[10];
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    s8 fr[4];
    for (int i = 0; i < 4; ++i) fr[i] = a;
    const int shift = STAG == 1 ? wave : 0;
    size_t off = ((size_t)blockIdx.x * 4 + wave) * 65536 % region;
    const char* lane_src = PAT == 0 ? src + lane * 16 : src + (size_t)(lane >> 3) * 2560 + (((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned lds_rd = (unsigned)(uintptr_t)(smem + wave * 16384 + lane * 16);
    int slot = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < periods; ++it) {
        if (STAG >= 2 && (it & 3) == 0) {      // a K tile of 64 = 4 periods: barrier, then wave w starts 16 w cycles late (SKEW) or not (BARRIER only)
            __builtin_amdgcn_s_barrier();
            if (STAG == 3) {
                if (wave >= 1) asm volatile("s_nop 15");
                if (wave >= 2) asm volatile("s_nop 15");
                if (wave >= 3) asm volatile("s_nop 15");
            }
        }
#pragma unroll
        for (int g = 0; g < 20; ++g) {
            mfma32(acc[g % 10], g & 1 ? fr[1] : a, g & 2 ? fr[2] : b);
            if (READS && (g & 1) == 0 && g < 18) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fr[(g >> 1) & 3]) : "v"(lds_rd), "n"((g >> 1) * 1024));   // "+v": the register stays live (a dead "=v" output is reused while the read is in flight)
            }
            if (READS && g == 19) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (ND > 0) {
                // gaps 1 + shift + 4 i, i < ND
                const int rel = g - 1;
                bool mine = false;
#pragma unroll
                for (int i = 0; i < ND; ++i) mine |= (rel == shift + 4 * i);
                if (g >= 1 && g <= 4 * (ND - 1) + 4 && mine) {
                    __builtin_amdgcn_global_load_lds((gptr_t)(lane_src + off), (lptr_t)(smem + 65536 + wave * 16384 + (slot & 15) * 1024), 16, 0, 0);
                    ++slot;
                    off += PAT == 0 ? 1024 : 8 * 2560;
                    if (off + 8 * 2560 + 1024 > region) off = 0;
                }
            }
        }
        if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (ND > 0 ? ND : 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 10; ++i) res += acc[i][0] + acc[i][15];
    for (int i = 0; i < 4; ++i) res += (float)fr[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool READS, int ND, int STAG, int PAT>
void run(const char* name, const char* src, size_t region) {
    float* out;
    long long* cyc;
    const int blocks = 256, periods = 2000;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 4 * 8);
    hipFuncSetAttribute((const void*)k<READS, ND, STAG, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<READS, ND, STAG, PAT>), dim3(blocks), dim3(256), 131072, 0, src, region, periods, out, cyc);
        hipDeviceSynchronize();
    }
    long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < 1024; ++i) { s += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
    const double per = s / 1024 / periods;
    printf("%-44s reads %d  DMA/period %d  %s  src %-10s: %7.1f cycles per period (20 MFMA = 640 ideal; slowest wave %.1f)  MFMA busy %.3f\n",
           name, (int)READS, ND, STAG == 0 ? "SAME gaps        " : STAG == 1 ? "STAGgered gaps  " : STAG == 2 ? "SAME + barrier/4 " : "SAME + barrier/4 + 16 w cycle skew", PAT ? "8x128B" : "1KiB", per, mx / periods, 640.0 / per);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    char* buf;
    const size_t big = (size_t)64 << 20;
    hipMalloc(&buf, big + (1 << 20));
    hipMemset(buf, 1, big + (1 << 20));
    const size_t l2 = (size_t)2 << 20;
    run<false, 0, 0, 0>("MFMA only", buf, l2);
    run<true, 0, 0, 0>("MFMA + fragment reads", buf, l2);
    run<true, 0, 2, 0>("MFMA + fragment reads", buf, l2);
#define FOUR(R, ND, PAT, NAME, BUF, REG) run<R, ND, 0, PAT>(NAME, BUF, REG); run<R, ND, 1, PAT>(NAME, BUF, REG); run<R, ND, 2, PAT>(NAME, BUF, REG); run<R, ND, 3, PAT>(NAME, BUF, REG);
    FOUR(false, 5, 0, "MFMA + DMA, L2-resident", buf, l2)
    FOUR(true, 4, 0, "MFMA + reads + DMA, L2-resident", buf, l2)
    FOUR(true, 5, 0, "MFMA + reads + DMA, L2-resident", buf, l2)
    FOUR(true, 5, 1, "MFMA + reads + DMA (A rows), L2-resident", buf, l2)
    FOUR(true, 5, 0, "MFMA + reads + DMA, 64 MiB streamed", buf, big)
    FOUR(true, 5, 1, "MFMA + reads + DMA (A rows), 64 MiB streamed", buf, big)
    return 0;
}
