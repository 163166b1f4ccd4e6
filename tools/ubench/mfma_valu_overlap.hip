// Do an MFMA stream of one wave and a VALU / LDS / VMEM stream of ANOTHER wave on the same SIMD overlap?
// A 512-thread block per CU: waves w and w + 4 share SIMD w (HW_ID, see tools/ffn_trace.py).  Waves 0..3 run role A, waves
// 4..7 role B; each role is timed with s_memtime alone (the other role exits at once) and together.
//   role A: 512 back-to-back v_mfma_f32_16x16x32_bf16 on 8 independent accumulators (16 cycles each when alone)
//   role B: one of  v_fma_f32 / v_pk_fma_f32 / v_cvt_pk_bf16_f32 / v_exp_f32 / ds_read_b128 / v_dot2c_f32_bf16 streams
// Question behind it (csrc/temporal.hip resident form, csrc/ffn.hip): how much of a VALU-bound epilogue of one wave can hide
// behind the MFMA main loop of its SIMD partner?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x

typedef float f16v __attribute__((ext_vector_type(16)));

// M32: role A issues v_mfma_f32_32x32x16_bf16 (same FLOPs per instruction-cycle, half the instructions) — round 4: does the
// partner get more issue slots per MFMA cycle next to the longer instruction?
template <int KIND, bool M32 = false>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, float seed, int run_a, int run_b) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed * i;
    __syncthreads();
    float res = 0.f;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (!run_a) return;
        f4 acc[8];
        s8 a, b;
        for (int i = 0; i < 8; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
        if (M32) {
            f16v acc2[4];
            for (int i = 0; i < 4; ++i)
                for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < 64; ++it) {   // 256 instructions = the FLOPs of 512 16x16x32
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2[i], 0, 0, 0);
            }
            t1 = __builtin_readcyclecounter();
            for (int i = 0; i < 4; ++i) res += acc2[i][0] + acc2[i][15];
        } else {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][3];
        }
    } else {
        if (!run_b) return;
        float x[8];
        f2 p[8];
        for (int i = 0; i < 8; ++i) { x[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{x[i], x[i] + 1.f}; }
        const f2 c2 = f2{seed, seed};
        const float* lp = lds + (threadIdx.x & 63) * 4;
        f4 l[8];
        for (int i = 0; i < 8; ++i) l[i] = f4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 64; ++it) {   // 64 x 8 = 512 instructions
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(seed));
            } else if (KIND == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                             "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));
            } else if (KIND == 2) {
                asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                             "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
            } else if (KIND == 3) {
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                             "v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
            } else if (KIND == 4) {   // 8 ds_read_b128, drained once per 8
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                             "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                             "s_waitcnt lgkmcnt(0)"
                             : "=v"(l[0]), "=v"(l[1]), "=v"(l[2]), "=v"(l[3]), "=v"(l[4]), "=v"(l[5]), "=v"(l[6]), "=v"(l[7])
                             : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)lp));
            } else if (KIND >= 10) {   // round 4: one- / two- / three-source VALU kinds (which classes does a busy matrix pipe slow down?)
#define OP8(txt) asm volatile(txt(0) txt(1) txt(2) txt(3) txt(4) txt(5) txt(6) txt(7) \
                              : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(seed), "v"(c2))
#define T_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define T_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define T_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define T_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %8\n"
#define T_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define T_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define T_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define T_MAD(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define T_FMAC(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define T_SUBREV(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
                if (KIND == 10) OP8(T_MUL); else if (KIND == 11) OP8(T_ADD); else if (KIND == 12) OP8(T_MAX);
                else if (KIND == 13) OP8(T_MAX3); else if (KIND == 14) OP8(T_ADDU); else if (KIND == 15) OP8(T_LSHL);
                else if (KIND == 16) OP8(T_MOV); else if (KIND == 17) OP8(T_AND); else if (KIND == 18) OP8(T_MAD);
                else if (KIND == 19) OP8(T_FMAC); else if (KIND == 20) OP8(T_SUBREV);
                else if (KIND == 21) {
                    asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                                 "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));
                } else if (KIND == 22) {
                    asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                                 "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));
                }
            } else if (KIND == 5) {
                asm volatile("v_dot2c_f32_bf16 %0, %8, %8\n v_dot2c_f32_bf16 %1, %8, %8\n v_dot2c_f32_bf16 %2, %8, %8\n v_dot2c_f32_bf16 %3, %8, %8\n"
                             "v_dot2c_f32_bf16 %4, %8, %8\n v_dot2c_f32_bf16 %5, %8, %8\n v_dot2c_f32_bf16 %6, %8, %8\n v_dot2c_f32_bf16 %7, %8, %8"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(seed));
            }
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 8; ++i) res += x[i] + p[i].x + p[i].y + l[i][0];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        if (wave == 0 || wave == 4) cyc[wave >> 2] = t1 - t0;
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cyc[2 + wave] = (hw >> 4) & 3;      // SIMD of this wave (round 4: the verdict depends on waves 0 and 4 sharing one)
    }
}

template <int KIND, bool M32 = false>
void run(const char* name) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 128);
    double r[3][2] = {};
    const int modes[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, 128);
            hipLaunchKernelGGL((k<KIND, M32>), dim3(256), dim3(512), 0, 0, out, cyc, 1.0001f, modes[m][0], modes[m][1]);
        }
        long long h[10] = {0};
        hipMemcpy(h, cyc, 80, hipMemcpyDeviceToHost);
        if (m == 2) printf("[simd of waves 0..7: %lld%lld%lld%lld %lld%lld%lld%lld] ", h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
        r[m][0] = h[0] / (M32 ? 256.0 : 512.0);
        r[m][1] = h[1] / 512.0;
    }
    printf("%-22s alone: MFMA %5.1f cyc/instr, B %5.1f cyc/instr | together: MFMA %5.1f, B %5.1f\n", name, r[0][0], r[1][1], r[2][0], r[2][1]);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("B = v_fma_f32");
    run<1>("B = v_pk_fma_f32");
    run<2>("B = v_cvt_pk_bf16_f32");
    run<3>("B = v_exp_f32");
    run<4>("B = ds_read_b128 x8+wait");
    run<5>("B = v_dot2c_f32_bf16");
    printf("-- round 4: which VALU classes does a busy matrix pipe slow down (role A = 16x16x32)\n");
    run<10>("B = v_mul_f32"); run<11>("B = v_add_f32"); run<12>("B = v_max_f32"); run<13>("B = v_max3_f32");
    run<14>("B = v_add_u32"); run<15>("B = v_lshlrev_b32"); run<16>("B = v_mov_b32"); run<17>("B = v_and_b32");
    run<18>("B = v_mad_u32_u24"); run<19>("B = v_fmac_f32"); run<20>("B = v_sub_f32"); run<21>("B = v_pk_mul_f32"); run<22>("B = v_pk_add_f32");
    printf("-- role A = v_mfma_f32_32x32x16_bf16 (4 accumulators)\n");
    run<0, true>("B = v_fma_f32");
    run<1, true>("B = v_pk_fma_f32");
    run<3, true>("B = v_exp_f32");
    run<4, true>("B = ds_read_b128 x8+wait");
    run<5, true>("B = v_dot2c_f32_bf16");
    return 0;
}
