// Round 5: how much VALU work hides behind MFMAs, by STRUCTURE (VERDICT r4 item 1 asks for the own-wave case first).
//   T1  one wave per SIMD: { MFMA ; N independent fillers } — cycles per MFMA vs N, for 32x32x16 and 16x16x32, fillers =
//       v_fma_f32 or a softmax-like mix (fma, exp, add, cvt_pk).  The guide's budget is <= 5 fillers per 32x32x16 gap.
//   T2  partner streams (mfma_valu_overlap.hip's A / B roles) with s_setprio on one of the roles.
//   T3  NW = 1 / 2 / 4 waves per SIMD, every wave running the SAME period { 16 MFMA 16x16x32 (or 8 32x32x16) + 64 VALU }, as
//       a burst pair (what hipcc emits for the attention tile: S^T burst, softmax, P.V burst) or hand-interleaved
//       (MFMA ; 4 VALU), with s_setprio 1 around the VALU burst / around the MFMA burst / never.  Reports the matrix pipe's
//       busy fraction of the SIMD = NW * MFMA cycles per period / period.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fill.hip -o /tmp/mfill && /tmp/mfill
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));

#define DEV __device__ __forceinline__

DEV void mfma16(f4& c, s8 a, s8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
DEV void mfma32(f16v& c, s8 a, s8 b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

// one filler instruction on chain i (8 independent chains); KIND 0: v_fma_f32, 1: softmax-like mix by position
template <int KIND>
DEV void filler(float (&x)[8], float s, int i, int pos) {
    float& r = x[i & 7];
    if (KIND == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(s));
    } else {
        switch (pos & 3) {
            case 0: asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(s)); break;
            case 1: asm volatile("v_exp_f32 %0, %0" : "+v"(r)); break;
            case 2: asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(s)); break;
            default: asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(s)); break;
        }
    }
}

// ---- T1
template <bool M32, int KIND, int N>
__global__ __launch_bounds__(256) void t1(float* out, long long* cyc, float seed) {
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f4 acc[8];
    f16v acc2[4];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        if (M32) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                mfma32(acc2[m], a, b);
#pragma unroll
                for (int f = 0; f < N; ++f) filler<KIND>(x, seed, m * N + f, m * N + f);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                mfma16(acc[m], a, b);
#pragma unroll
                for (int f = 0; f < N; ++f) filler<KIND>(x, seed, m * N + f, m * N + f);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += x[i] + acc[i][0];
    for (int i = 0; i < 4; ++i) res += acc2[i][0] + acc2[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <bool M32, int KIND, int N>
double run_t1() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((t1<M32, KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, 1.0001f);
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(out);
    hipFree(cyc);
    return (double)h / (M32 ? 256.0 : 512.0);
}

template <bool M32, int KIND>
void sweep_t1(const char* name) {
    printf("T1 %-34s cycles per MFMA at N fillers/MFMA  N=0: %5.1f  1: %5.1f  2: %5.1f  3: %5.1f  4: %5.1f  5: %5.1f  6: %5.1f  8: %5.1f  10: %5.1f\n",
           name, run_t1<M32, KIND, 0>(), run_t1<M32, KIND, 1>(), run_t1<M32, KIND, 2>(), run_t1<M32, KIND, 3>(), run_t1<M32, KIND, 4>(),
           run_t1<M32, KIND, 5>(), run_t1<M32, KIND, 6>(), run_t1<M32, KIND, 8>(), run_t1<M32, KIND, 10>());
}

// ---- T4 (round 5, after ffn32): as T1 with the MFMAs on NACC accumulators in turn — dependency distance NACC.  csrc/ffn32.hip runs
// its up-projection on TWO accumulators (V, G) and measured MFMA, VALU, DMA and LDS time adding up serially.
template <int NACC, int KIND, int N>
__global__ __launch_bounds__(256) void t4(float* out, long long* cyc, float seed) {
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f16v acc2[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            mfma32(acc2[m % NACC], a, b);
#pragma unroll
            for (int f = 0; f < N; ++f) filler<KIND>(x, seed, m * N + f, m * N + f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += x[i];
    for (int i = 0; i < 4; ++i) res += acc2[i][0] + acc2[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int KIND, int N>
double run_t4() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((t4<NACC, KIND, N>), dim3(256), dim3(256), 0, 0, out, cyc, 1.0001f);
    long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(out);
    hipFree(cyc);
    return (double)h / 256.0;
}
template <int NACC, int KIND>
void sweep_t4(const char* name) {
    printf("T4 %-44s cycles per MFMA at N fillers/MFMA  N=0: %5.1f  2: %5.1f  4: %5.1f  6: %5.1f  8: %5.1f\n", name, run_t4<NACC, KIND, 0>(),
           run_t4<NACC, KIND, 2>(), run_t4<NACC, KIND, 4>(), run_t4<NACC, KIND, 6>(), run_t4<NACC, KIND, 8>());
}

// ---- T2: role A (waves 0..3) MFMA stream, role B (waves 4..7) v_fma stream, priorities pa / pb
template <bool M32>
__global__ __launch_bounds__(512) void t2(float* out, long long* cyc, float seed, int run_a, int run_b, int pa, int pb) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f4 acc[8];
    f16v acc2[4];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    long long t0 = 0, t1 = 0;
    __syncthreads();
    if (wave < 4) {
        if (!run_a) return;
        if (pa == 3) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 64; ++it) {
            if (M32) {
#pragma unroll
                for (int m = 0; m < 4; ++m) mfma32(acc2[m], a, b);
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) mfma16(acc[m], a, b);
            }
        }
        t1 = __builtin_readcyclecounter();
    } else {
        if (!run_b) return;
        if (pb == 3) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int f = 0; f < 8; ++f) filler<1>(x, seed, f, f);
        }
        t1 = __builtin_readcyclecounter();
    }
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += x[i] + acc[i][0];
    for (int i = 0; i < 4; ++i) res += acc2[i][0] + acc2[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 4)) cyc[wave >> 2] = t1 - t0;
}

template <bool M32>
void run_t2(const char* name, int pa, int pb) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    double r[3][2] = {};
    const int modes[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL((t2<M32>), dim3(256), dim3(512), 0, 0, out, cyc, 1.0001f, modes[m][0], modes[m][1], pa, pb);
        }
        long long h[2] = {0, 0};
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        r[m][0] = h[0] / (M32 ? 256.0 : 512.0);
        r[m][1] = h[1] / 512.0;
    }
    printf("T2 %-20s prio A=%d B=%d  alone: MFMA %5.1f cyc, B(mix) %5.1f cyc/instr | together: MFMA %5.1f, B %5.1f\n", name, pa, pb, r[0][0],
           r[1][1], r[2][0], r[2][1]);
    hipFree(out);
    hipFree(cyc);
}

// ---- T3: NW waves per SIMD, each the same { MFMA work + 64 VALU } period.  INTER: 0 burst pair, 1 interleaved.
// PRIO: 0 none, 1 s_setprio 1 around the VALU burst, 2 around the MFMA burst, 3 (INTER) static prio by wave group.
template <bool M32, int INTER, int PRIO, int KIND>
__global__ __launch_bounds__(1024) void t3(float* out, long long* cyc, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
    s8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f4 acc[8];
    f16v acc2[4];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    __syncthreads();
    for (int g = 0; g < grp; ++g) {   // stagger the groups by a third of a period
#pragma unroll
        for (int f = 0; f < 24; ++f) filler<0>(x, seed, f, f);
    }
    if (PRIO == 3 && grp != 0) __builtin_amdgcn_s_setprio(1);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 48; ++it) {
        if (INTER == 0) {
            if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
            if (M32) {
#pragma unroll
                for (int m = 0; m < 8; ++m) mfma32(acc2[m & 3], a, b);
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) mfma16(acc[m & 7], a, b);
            }
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int f = 0; f < 64; ++f) filler<KIND>(x, seed, f, f);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        } else {
            if (M32) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    mfma32(acc2[m & 3], a, b);
#pragma unroll
                    for (int f = 0; f < 8; ++f) filler<KIND>(x, seed, m * 8 + f, m * 8 + f);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    mfma16(acc[m & 7], a, b);
#pragma unroll
                    for (int f = 0; f < 4; ++f) filler<KIND>(x, seed, m * 4 + f, m * 4 + f);
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float res = 0.f;
    for (int i = 0; i < 8; ++i) res += x[i] + acc[i][0];
    for (int i = 0; i < 4; ++i) res += acc2[i][0] + acc2[i][15];
    out[blockIdx.x * 1024 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave & 3) == 0) cyc[grp] = t1 - t0;
}

template <bool M32, int INTER, int PRIO, int KIND>
void run_t3(const char* name) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 64);
    printf("T3 %-58s", name);
    for (int nw = 1; nw <= 4; nw *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL((t3<M32, INTER, PRIO, KIND>), dim3(256), dim3(256 * nw), 0, 0, out, cyc, 1.0001f);
        }
        long long h[4] = {0, 0, 0, 0};
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int g = 0; g < nw; ++g) worst = h[g] > worst ? (double)h[g] : worst;
        const double period = worst / 48.0;
        printf("  NW=%d: period %6.0f cyc, pipe busy %.2f |", nw, period, nw * 256.0 / period);   // 16 x 16 = 8 x 32 = 256 MFMA cycles
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    sweep_t1<true, 0>("32x32x16 + v_fma_f32");
    sweep_t1<true, 1>("32x32x16 + mix(fma,exp,add,cvt)");
    sweep_t1<false, 0>("16x16x32 + v_fma_f32");
    sweep_t1<false, 1>("16x16x32 + mix(fma,exp,add,cvt)");
    sweep_t4<1, 0>("32x32x16 on ONE accumulator + v_fma_f32");
    sweep_t4<2, 0>("32x32x16 on TWO accumulators + v_fma_f32");
    sweep_t4<2, 1>("32x32x16 on TWO accumulators + mix");
    sweep_t4<4, 1>("32x32x16 on FOUR accumulators + mix");
    run_t2<false>("A = 16x16x32", 0, 0);
    run_t2<false>("A = 16x16x32", 0, 3);
    run_t2<false>("A = 16x16x32", 3, 0);
    run_t2<true>("A = 32x32x16", 0, 0);
    run_t2<true>("A = 32x32x16", 0, 3);
    run_t2<true>("A = 32x32x16", 3, 0);
    run_t3<false, 0, 0, 1>("16x16x32 burst(16 MFMA | 64 mix), no prio");
    run_t3<false, 0, 1, 1>("16x16x32 burst, setprio 1 around VALU");
    run_t3<false, 0, 2, 1>("16x16x32 burst, setprio 1 around MFMA");
    run_t3<false, 1, 0, 1>("16x16x32 interleaved (MFMA ; 4 mix)");
    run_t3<false, 1, 3, 1>("16x16x32 interleaved, younger groups prio 1");
    run_t3<true, 0, 0, 1>("32x32x16 burst(8 MFMA | 64 mix), no prio");
    run_t3<true, 0, 1, 1>("32x32x16 burst, setprio 1 around VALU");
    run_t3<true, 0, 2, 1>("32x32x16 burst, setprio 1 around MFMA");
    run_t3<true, 1, 0, 1>("32x32x16 interleaved (MFMA ; 8 mix)");
    run_t3<true, 1, 3, 1>("32x32x16 interleaved, younger groups prio 1");
    run_t3<true, 0, 0, 0>("32x32x16 burst(8 MFMA | 64 fma), no prio");
    run_t3<true, 1, 0, 0>("32x32x16 interleaved (MFMA ; 8 fma)");
    return 0;
}
