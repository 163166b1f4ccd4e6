// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wiw_svd.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one 16-byte MFMA operand
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef uint16_t bf16_t;

#define WIW_DEV __device__ __forceinline__

// ---- 16-bit storage type of THIS build.  The library is compiled twice from the same sources: bf16 (libwiwsvd.so,
// the default) and, with -DWIW_F16, IEEE fp16 (libwiwsvd_f16.so: the reference's served default dtype,
// FTsvd/eval_inference.py:294).  Everything dtype-specific is here: the conversions, the MFMA opcode and the constant 1.0;
// the kernels move 16-bit words and accumulate in fp32 either way.  (Helper names keep their "bf" spelling.)
typedef float wiw_f32x2 __attribute__((ext_vector_type(2)));
#ifdef WIW_F16
typedef _Float16 wiw_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wiw_h16x8 __attribute__((ext_vector_type(8)));
#define WIW_DTYPE_CODE 1
#define WIW_ONE16 0x3C00   /* 1.0 */
WIW_DEV float bf2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
WIW_DEV wiw_f32x2 unpack2(uint32_t u) { return __builtin_convertvector(__builtin_bit_cast(wiw_h16x2, u), wiw_f32x2); }
// float -> fp16, round-to-nearest-even (v_cvt_f16_f32 x2 + v_pack)
WIW_DEV uint32_t pack2bf(float lo, float hi) {
    const wiw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wiw_h16x2));
}
#define WIW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wiw_h16x8, a), __builtin_bit_cast(wiw_h16x8, b), c, 0, 0, 0)
#else
typedef __bf16 wiw_bf16x2 __attribute__((ext_vector_type(2)));
#define WIW_DTYPE_CODE 0
#define WIW_ONE16 0x3F80   /* 1.0 */
WIW_DEV float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
WIW_DEV wiw_f32x2 unpack2(uint32_t u) { return wiw_f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
// float -> bf16, round-to-nearest-even, on the native gfx950 converter (v_cvt_pk_bf16_f32: 2 values/instr)
WIW_DEV uint32_t pack2bf(float lo, float hi) {
    const wiw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wiw_bf16x2));
}
#define WIW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
// c + a.lo*b.lo + a.hi*b.hi on one packed 16-bit pair in ONE VALU instruction (v_dot2c_f32_bf16 / v_dot2c_f32_f16, fp32
// products and sum) — row statistics straight from MFMA operand registers, no unpack
WIW_DEV float dot2_acc(uint32_t a, uint32_t b, float c) {
#ifdef WIW_F16
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(wiw_h16x2, a), __builtin_bit_cast(wiw_h16x2, b), c, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(wiw_bf16x2, a), __builtin_bit_cast(wiw_bf16x2, b), c, false);
#endif
}
WIW_DEV uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.0f) & 0xffffu); }
WIW_DEV float silu_f(float x) {   // x * sigmoid(x) on the raw v_exp_f32 / v_rcp_f32 (1 ulp each; output is bf16)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// erf-GELU (activations.py:109 `F.gelu`, approximate='none'): x * Phi(x).
// Round 5: Phi(x) = sigmoid(x * P(min(|x|, 6))) with P of degree 4 fitted (minimax on the absolute error of
// x * Phi(x) over the whole line, /tmp-free recipe in oracle/gelu_fit.py) — |error| <= 5.2e-6 evaluated in fp32 with the raw
// v_exp_f32 / v_rcp_f32, i.e. 1/50 of half an fp16 ulp at |h| = 1 and 1/400 of bf16's; 10 VALU instructions instead of the
// 14 of Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), which stays as gelu_erf_as_f: the fine-tuning kernels (train.hip) keep
// it, and -DWIW_GELU_AS7126=1 builds every epilogue with it (the A/B switch).  The GEGLU epilogues evaluate 20480 of these
// per 256x160 block tile and their VALU issue time is NOT hidden behind MFMAs.  The coefficients carry -log2(e).
#ifndef WIW_GELU_AS7126
#define WIW_GELU_AS7126 0
#endif
#define WIW_GELU_C0 (-2.3031814098358154f)
#define WIW_GELU_C1 (0.0026162799913436174f)
#define WIW_GELU_C2 (-0.1058432012796402f)
#define WIW_GELU_C3 (-0.0018446178874000907f)
#define WIW_GELU_C4 (0.0014851129380986094f)
#define WIW_GELU_CLAMP 6.0f
WIW_DEV float gelu_erf_as_f(float x) {
    // 14 VALU instructions: v_rcp_f32 / v_exp_f32 raw (1 ulp is far below the bf16 output rounding),
    // 0.5*x*(1 + sign(x)*erf|x|) rewritten as 0.5*(x + |x|*erf|x|)
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = 1.061405429f;
    poly = __builtin_fmaf(poly, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = __builtin_fmaf(-poly * t, e, 1.0f);
    return 0.5f * __builtin_fmaf(ax, erf_abs, x);
}
WIW_DEV float gelu_erf_f(float x) {
#if WIW_GELU_AS7126
    return gelu_erf_as_f(x);
#else
    const float a = __builtin_fminf(fabsf(x), WIW_GELU_CLAMP);     // beyond 6: x * sigmoid(3.3 x) is x or 0 to fp32 roundoff
    float p = __builtin_fmaf(WIW_GELU_C4, a, WIW_GELU_C3);
    p = __builtin_fmaf(p, a, WIW_GELU_C2);
    p = __builtin_fmaf(p, a, WIW_GELU_C1);
    p = __builtin_fmaf(p, a, WIW_GELU_C0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));     // exp2 -> inf for x << 0: rcp(inf) = 0
#endif
}

// gelu_erf_f on NP PACKED pairs at once, breadth first, the same arithmetic bit for bit (both forms).  One wave issues at most one VALU
// instruction per ~5.4 cycles (8.5 when it depends on the previous one) packed or not (tools/ubench/valu_rate.hip), so the
// cost of a GELU block is its instruction count: v_pk_* halves it and independent work per stage removes the dependency
// stalls.  wiw_regp is an empty asm statement that reads and writes every pair of a stage — without it instruction
// selection re-serialises the chains value by value (csrc/ffn.hip).
template <int NP>
WIW_DEV void wiw_regp(wiw_f32x2 (&a)[NP]) {
    static_assert(NP == 4 || NP == 5 || NP == 8 || NP == 10, "add the operand list for this width");
    if constexpr (NP == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    if constexpr (NP == 5) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
    if constexpr (NP == 8)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
    if constexpr (NP == 10)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                     "+v"(a[8]), "+v"(a[9]));
}
template <int NP>
WIW_DEV void gelu_erf_pk(wiw_f32x2 (&x)[NP]) {
#if !WIW_GELU_AS7126
    wiw_f32x2 a[NP], p[NP];
    const auto fma2s = [](wiw_f32x2 a_, wiw_f32x2 b_, wiw_f32x2 c_) { return __builtin_elementwise_fma(a_, b_, c_); };
    const auto bcs = [](float c) { return wiw_f32x2{c, c}; };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        a[i] = wiw_f32x2{__builtin_fminf(fabsf(x[i].x), WIW_GELU_CLAMP), __builtin_fminf(fabsf(x[i].y), WIW_GELU_CLAMP)};
        p[i] = fma2s(bcs(WIW_GELU_C4), a[i], bcs(WIW_GELU_C3));
    }
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = fma2s(p[i], a[i], bcs(WIW_GELU_C2));
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = fma2s(p[i], a[i], bcs(WIW_GELU_C1));
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = fma2s(p[i], a[i], bcs(WIW_GELU_C0));
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = x[i] * p[i];
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = wiw_f32x2{__builtin_amdgcn_exp2f(p[i].x), __builtin_amdgcn_exp2f(p[i].y)};
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = p[i] + bcs(1.0f);
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = wiw_f32x2{__builtin_amdgcn_rcpf(p[i].x), __builtin_amdgcn_rcpf(p[i].y)};
    wiw_regp(p);
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = x[i] * p[i];
    wiw_regp(x);
    return;
#endif

    wiw_f32x2 ax[NP], z[NP], t[NP], e[NP], poly[NP];
    const auto fma2 = [](wiw_f32x2 a, wiw_f32x2 b, wiw_f32x2 c) { return __builtin_elementwise_fma(a, b, c); };
    const auto bc = [](float c) { return wiw_f32x2{c, c}; };
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        ax[i] = wiw_f32x2{fabsf(x[i].x), fabsf(x[i].y)};
        z[i] = ax[i] * 0.70710678118654752f;
    }
    wiw_regp(z);
#pragma unroll
    for (int i = 0; i < NP; ++i) { t[i] = fma2(bc(0.3275911f), z[i], bc(1.0f)); e[i] = (z[i] * -1.4426950408889634f) * z[i]; }
    wiw_regp(t); wiw_regp(e);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        t[i] = wiw_f32x2{__builtin_amdgcn_rcpf(t[i].x), __builtin_amdgcn_rcpf(t[i].y)};
        e[i] = wiw_f32x2{__builtin_amdgcn_exp2f(e[i].x), __builtin_amdgcn_exp2f(e[i].y)};
    }
    wiw_regp(t); wiw_regp(e);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = fma2(bc(1.061405429f), t[i], bc(-1.453152027f));
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = fma2(poly[i], t[i], bc(1.421413741f));
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = fma2(poly[i], t[i], bc(-0.284496736f));
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = fma2(poly[i], t[i], bc(0.254829592f));
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = -poly[i] * t[i];
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) poly[i] = fma2(poly[i], e[i], bc(1.0f));     // erf|x|
    wiw_regp(poly);
#pragma unroll
    for (int i = 0; i < NP; ++i) x[i] = fma2(ax[i], poly[i], x[i]) * 0.5f;
    wiw_regp(x);
}

WIW_DEV void unpack8(const uint4& v, float* f) {
    const wiw_f32x2 a = unpack2(v.x), b = unpack2(v.y), c = unpack2(v.z), d = unpack2(v.w);
    f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = c.x; f[5] = c.y; f[6] = d.x; f[7] = d.y;
}
WIW_DEV uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// Cross-lane exchange WITHOUT the LDS crossbar (`__shfl_xor` compiles to ds_bpermute_b32: an LDS-pipe instruction
// with > 100 cycles of latency; with two dependent ones per row a C = 320 LayerNorm was shuffle-bound at 2.7 TB/s):
//   * within a row of 16 lanes: DPP row rotations folded into the VALU add (v_add_f32_dpp row_ror:n);
//   * lane ^ 16 / lane ^ 32: gfx950's v_permlane16_swap / v_permlane32_swap (VALU).  With both operands = x the two
//     results hold x of "my" half and x of the partner half for every lane.
template <int CTRL>
WIW_DEV float dpp_mov(float x) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
WIW_DEV float xor16_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
WIW_DEV float xor32_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
WIW_DEV float xor16_max(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
WIW_DEV float xor32_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// wave-wide (64-lane) reductions; every lane receives the result
WIW_DEV float wave_sum(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1  -> every lane holds the sum of its row of 16
    return xor32_sum(xor16_sum(v));
}
WIW_DEV float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x122>(v));
    v = fmaxf(v, dpp_mov<0x121>(v));
    return xor32_max(xor16_max(v));
}

// Host-side error bookkeeping (elementwise.hip)
void wiw_set_error(const char* msg);
#define WIW_REQUIRE(cond, msg)        \
    do {                              \
        if (!(cond)) {                \
            wiw_set_error(msg);       \
            return WIW_EINVAL;        \
        }                             \
    } while (0)
int wiw_check_launch(const char* what);
