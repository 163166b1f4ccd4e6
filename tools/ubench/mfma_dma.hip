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

// ---- 8 waves per block (2 per SIMD), the 64x160 wave tile of gemm_huge.hip on 32x32x16 MFMAs: one 16-k step = 10 MFMA + 7 ds_read_b128
// (2 A + 5 W fragments), 9 DMA pieces per wave and K tile of 64 (4 periods), one s_barrier per K tile.  LDS traffic of the CU per
// K tile: 8 waves x 28 KiB of reads + 72 KiB of fill in 2560 MFMA cycles per SIMD = 87.5 + 28 B/clk.
template <bool READS, int DMA, bool BAR, int RD>
__global__ __launch_bounds__(512) void k8(const char* src, size_t region, int periods, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f16v acc[10];
    for (int i = 0; i < 10; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    s8 fr[8];
    for (int i = 0; i < 8; ++i) fr[i] = a;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 65536 % region;
    const char* lane_src = src + lane * 16;
    // fragment reads: conflict-free image (16-byte chunk XOR-swizzled by the row), waves read different regions of a 72 KiB stage
    const unsigned lds_rd = (unsigned)(uintptr_t)(smem + (wave & 3) * 8192 + (lane & 31) * 64 + ((((lane >> 5) ^ ((lane >> 1) & 3)) & 3) << 4));
    int slot = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < periods; ++it) {
        if (BAR && (it & 3) == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            mfma32(acc[g], g & 1 ? fr[1] : fr[0], fr[2 + (g >> 1)]);
            if (READS && g < RD) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fr[g]) : "v"(lds_rd), "n"(g * 2048));
            if (READS && g == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // DMA 1: every wave in gaps 2, 6 (+ 8 once per K tile); 2: the second wave of a SIMD (waves 4..7) in gaps 4, 8 (+ 0) instead;
            // 3: like 2, and 4 of the 9 pieces read eight 128-byte row segments (an A operand)
            const bool hi = DMA >= 2 && wave >= 4;
            const bool now = hi ? (g == 4 || g == 8 || (g == 0 && (it & 3) == 2)) : (g == 2 || g == 6 || (g == 8 && (it & 3) == 0));
            if (DMA && now) {      // 2 + 2 + 2 + 3 = 9 per 4 periods
                const char* g_src = (DMA == 3 && (slot & 1)) ? src + off + (size_t)(lane >> 3) * 2560 + (((lane & 7) ^ (lane >> 3)) << 4) : lane_src + off;
                __builtin_amdgcn_global_load_lds((gptr_t)g_src, (lptr_t)(smem + 73728 + wave * 8192 + (slot & 7) * 1024), 16, 0, 0);
                ++slot;
                off += 1024;
                if (off + 8 * 2560 + 2048 > region) off = 0;
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 10; ++i) res += acc[i][0] + acc[i][15];
    for (int i = 0; i < 8; ++i) res += (float)fr[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool READS, int DMA, bool BAR, int RD>
void run8(const char* name, const char* src, size_t region) {
    float* out;
    long long* cyc;
    const int blocks = 256, periods = 4000;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipFuncSetAttribute((const void*)k8<READS, DMA, BAR, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k8<READS, DMA, BAR, RD>), dim3(blocks), dim3(512), 147456, 0, src, region, periods, out, cyc);
        hipDeviceSynchronize();
    }
    long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < 2048; ++i) { s += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
    const double per = s / 2048 / periods;      // per wave period; the SIMD runs two waves: ideal 640 cycles per pair of periods
    printf("8 waves: %-40s reads/step %d  DMA form %d  barrier/K-tile %d: %7.1f cycles per wave period (2 waves x 10 MFMA = 640 ideal; slowest %.1f)  MFMA busy %.3f\n",
           name, READS ? RD : 0, (int)DMA, (int)BAR, per, mx / periods, 640.0 / per);
    hipFree(out);
    hipFree(cyc);
}

// ---- the same on 16x16x32 MFMAs (the instruction gemm_huge.hip's accumulator layout and epilogues are built on): one 32-k step of the
// 64x160 wave tile = 40 MFMA (16 cycles each) + 14 ds_read_b128 (4 A + 10 W fragments), 9 DMA pieces per 2 periods (a K tile of 64)
typedef float f4 __attribute__((ext_vector_type(4)));
DEV void mfma16(f4& c, s8 a, s8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
template <bool READS, int DMA, bool BAR>
__global__ __launch_bounds__(512) void k16(const char* src, size_t region, int periods, float* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f4 acc[4][10];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 10; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    s8 fa[2], fb[10];
    fa[0] = fa[1] = a;
    for (int i = 0; i < 10; ++i) fb[i] = b;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 65536 % region;
    const char* lane_src = src + lane * 16;
    const unsigned lds_rd = (unsigned)(uintptr_t)(smem + (wave & 3) * 8192 + (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7)) & 7) << 4));
    int slot = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < periods; ++it) {
        if (BAR && (it & 1) == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 10; ++ni) {
                mfma16(acc[mi][ni], fa[mi & 1], fb[ni]);
                if (READS) {
                    // rolling W reload behind the last A fragment's MFMA that used it; the next A fragment one per mi
                    if (mi == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fb[ni]) : "v"(lds_rd), "n"(ni * 2048));
                    if (ni == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fa[(mi + 1) & 1]) : "v"(lds_rd), "n"(32768 + mi * 2048));
                    if (mi == 3 && ni == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                const int g = mi * 10 + ni;
                const bool now = (g % 8 == 3 && g < 32) || (g == 36 && (it & 1) == 0);      // 4 + 4 + 1 = 9 per 2 periods
                if (DMA && now) {
                    const char* g_src = (DMA == 3 && (slot & 1)) ? src + off + (size_t)(lane >> 3) * 2560 + (((lane & 7) ^ (lane >> 3)) << 4) : lane_src + off;
                    __builtin_amdgcn_global_load_lds((gptr_t)g_src, (lptr_t)(smem + 73728 + wave * 8192 + (slot & 7) * 1024), 16, 0, 0);
                    ++slot;
                    off += 1024;
                    if (off + 8 * 2560 + 2048 > region) off = 0;
                }
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 10; ++j) res += acc[i][j][0] + acc[i][j][3];
    for (int i = 0; i < 10; ++i) res += (float)fb[i][0];
    res += (float)fa[0][0] + (float)fa[1][0];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <bool READS, int DMA, bool BAR>
void run16(const char* name, const char* src, size_t region) {
    float* out;
    long long* cyc;
    const int blocks = 256, periods = 4000;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipFuncSetAttribute((const void*)k16<READS, DMA, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k16<READS, DMA, BAR>), dim3(blocks), dim3(512), 147456, 0, src, region, periods, out, cyc);
        hipDeviceSynchronize();
    }
    long long h[2048];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, mx = 0;
    for (int i = 0; i < 2048; ++i) { s += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
    const double per = s / 2048 / periods;
    printf("8 waves, 16x16x32: %-44s reads %d  DMA form %d  barrier/K-tile %d: %7.1f cycles per wave period (2 waves x 40 MFMA = 1280 ideal; slowest %.1f)  MFMA busy %.3f\n",
           name, (int)READS, DMA, (int)BAR, per, mx / periods, 1280.0 / per);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
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
    run8<false, 0, false, 7>("MFMA only", buf, l2);
    run8<true, 0, false, 7>("MFMA + fragment reads", buf, l2);
    run8<true, 0, false, 10>("MFMA + fragment reads (10 per step)", buf, l2);
    run8<true, 0, true, 7>("MFMA + reads + barrier", buf, l2);
    run8<false, 1, false, 7>("MFMA + DMA", buf, l2);
    run8<true, 1, false, 7>("MFMA + reads + DMA", buf, l2);
    run8<true, 1, true, 7>("MFMA + reads + DMA + barrier", buf, l2);
    run8<true, 2, false, 7>("MFMA + reads + DMA (SIMD partners in different gaps)", buf, l2);
    run8<true, 2, true, 7>("MFMA + reads + DMA (partners differ) + barrier", buf, l2);
    run8<true, 3, true, 7>("... + 4 of 9 pieces as A row segments", buf, l2);
    run8<true, 3, true, 7>("... the same, 64 MiB source", buf, big);
    run16<false, 0, false>("MFMA only", buf, l2);
    run16<true, 0, false>("MFMA + fragment reads", buf, l2);
    run16<true, 0, true>("MFMA + reads + barrier", buf, l2);
    run16<true, 1, false>("MFMA + reads + DMA", buf, l2);
    run16<true, 1, true>("MFMA + reads + DMA + barrier", buf, l2);
    run16<true, 3, true>("... + 4 of 9 pieces as A row segments", buf, l2);
    run16<true, 3, true>("... the same, 64 MiB source", buf, big);
    return 0;
}
