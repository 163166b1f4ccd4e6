// Spatial flash attention, second form (round 5): v_mfma_f32_32x32x16, head_dim 64, the softmax of one half tile issued
// BETWEEN the matrix instructions of the neighbouring half tile.
//
// Why a second form (tools/ubench/mfma_fill.hip, profiles/r11a_mfma_fill_ubench.txt): a wave that issues its MFMAs as a burst
// and its softmax as a block of VALU instructions keeps the matrix pipe 0.59-0.63 busy even with four such waves on a SIMD
// (a stalled MFMA holds the issue port against the other waves' VALU); the SAME instructions in the order
// { MFMA ; 4-8 VALU } reach 0.71-0.72, and a 32x32x16 MFMA hides 4 own-wave VALU issues completely (5: +6 cycles) where a
// 16x16x32 hides one.  The first form (attention.hip) is the burst order, as hipcc schedules it: 0.50 busy at S = 9216.
//
// Formulation (transposed, as the first form — probabilities never leave the registers, no operand is transposed in LDS):
//     S'^T = K . Q^T - m      A = K rows (32 keys x 16 d per MFMA), B = Q^T, C = -m (the running reference of the lane's query)
//     O^T += V^T . P^T        A = V^T rows (32 d x 16 keys),        B = P^T straight from the S'^T accumulator registers
// * Q arrives PRE-SCALED by log2(e) / sqrt(64): the factor is folded into the to_q rows of the q|k|v projection weights when
//   they are packed (unet.py), before their one rounding to 16 bits.  With the reference -m in the MFMA's C operand the
//   accumulator holds the exponent itself: p = v_exp_f32(s') is the only instruction per score besides its share of the
//   maximum, the row sum and the 16-bit pack (2.9 VALU per score; the first form: 3.9).
// * C/D layout of 32x32x16: lane (n = lane & 31, hi = lane >> 5), register r <-> row (r & 3) + 8 (r >> 2) + 4 hi.  The K row
//   that feeds MFMA row i is free: key(i) = 16 (i >> 4) + 8 ((i >> 2) & 1) + 4 ((i >> 3) & 1) + (i & 3) makes registers
//   8 ks .. 8 ks + 7 of lane (q, hi) the EIGHT CONSECUTIVE keys 16 ks + 8 hi .. + 7: exactly the B operand (k = 8 hi + e) of
//   the P.V MFMA of key step ks — no cross-lane traffic (the usual layout needs 8 v_permlane32_swap per 32 scores), and
//   the V^T A operand is one aligned 16-byte LDS read.
// * Lazy reference: m only has to bound the exponents.  It is the exact row maximum of the first 32 keys and is raised (O, l,
//   the pending scores rescaled) only when some query of the wave exceeds it by more than 2^8; the test costs 8 v_max3 and one
//   compare per 16 scores, the cross-lane part runs in the rare branch only.  Branches sit at segment boundaries where
//   everything at the old scale (O, l, the undecided score block) is rescaled exactly once (cdna guide T13).
// * Schedule per wave and 64-key tile j (half tiles h0, h1 of 32 keys; four MFMAs of 32 cycles per segment):
//       A: S0 = K(j,h0).Q      between them: exp / sum / pack of S1's key step 0, LDS reads of V(j-1,h1)
//       B: O += V(j-1,h1).P1   between them: exp / sum / pack of S1's key step 1, max of S0, LDS reads of K(j,h1)
//       C: S1 = K(j,h1).Q      ...of S0's key step 0, LDS reads of V(j,h0);   then wait for DMA(j+1), ONE barrier, issue DMA(j+2)
//       D: O += V(j,h0).P0     ...of S0's key step 1, max of S1, LDS reads of K(j+1,h0)
//   K / V^T tiles arrive by LDS-DMA into a ring of THREE 16-KiB stages (tile j-1's V^T is still read in segment A of tile j).
// * LDS images: K tile [64 keys][128 B], V^T tile [64 d][128 B]; the 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 7):
//   every 16-lane group of a ds_read_b128 (rows distinct mod 16, same chunk) covers all 64 banks once.
#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
WIW_DEV void glds16(const char* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

#ifdef WIW_F16
#define WIW_MFMA32(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wiw_h16x8, a), __builtin_bit_cast(wiw_h16x8, b), c, 0, 0, 0)
#else
#define WIW_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

#ifndef WIW_A32_SCHED
#define WIW_A32_SCHED 1   // 1: sched_group_barrier interleave inside the segments; 0: hipcc's own order (A/B)
#endif
#ifndef WIW_A32_VALU_A
#define WIW_A32_VALU_A 6   // VALU issues requested per MFMA gap in segments A / C
#endif
#ifndef WIW_A32_VALU_B
#define WIW_A32_VALU_B 8   // ... in segments B / D
#endif

WIW_DEV float max3r(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
WIW_DEV float max2r(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The LDS reads of a segment feed the NEXT segment's MFMAs; LLVM's IR-level sinking would move them next to their users
// (sched_barrier only fences the machine scheduler).  An empty asm that reads the fragments keeps them in their segment.
WIW_DEV void pin4(const bf16x8 (&f)[4]) { asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3])); }
WIW_DEV float max16(const f32x16& x) {   // 8 instructions, depth 3
    const float a = max3r(x[0], x[1], x[2]), b = max3r(x[3], x[4], x[5]), c = max3r(x[6], x[7], x[8]);
    const float d = max3r(x[9], x[10], x[11]), e = max3r(x[12], x[13], x[14]);
    return max2r(max3r(a, b, x[15]), max3r(c, d, e));
}
WIW_DEV float partner_max(float x) {   // max over the two lanes (hi = 0, 1) of a query
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max2r(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

constexpr int A2_STAGE = 16384, A2_NSTAGE = 3;
constexpr float A2_THR = 8.0f;

// mask values of __builtin_amdgcn_sched_group_barrier (LLVM SchedGroupMask)
#define SG_VALU 0x402   /* VALU | TRANS */
#define SG_MFMA 0x8
#define SG_DSR 0x100
#if WIW_A32_SCHED
#define A32_INTERLEAVE(NV)                                          \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {              \
        __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);        \
        __builtin_amdgcn_sched_group_barrier(SG_DSR, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier(SG_VALU, NV, 0);       \
    }
#else
#define A32_INTERLEAVE(NV)
#endif

__global__ __launch_bounds__(256, 3) void attn_spatial32_kernel(const uint16_t* __restrict__ QK, int ldqk, int k_col_off,
                                                                const uint16_t* __restrict__ Vt, int64_t ldvt,
                                                                uint16_t* __restrict__ O, int ldo, int S, int heads, int q_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[A2_NSTAGE * A2_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware remap: an XCD walks a contiguous range of (frame, head, q_tile), K / V^T stay in its L2
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int qt = bid % q_tiles;
    const int fh = bid / q_tiles;
    const int h = fh % heads, n = fh / heads;
    const int64_t row0 = (int64_t)n * S;
    const int li = lane & 31, hi = lane >> 5;
    const int qrow = qt * 128 + wave * 32 + li;

    // ---- Q^T B operands (lane: query li, d = 16 ks + 8 hi .. + 7), pre-scaled by the projection
    bf16x8 qf[4];
    {
        const uint16_t* src = QK + (row0 + qrow) * ldqk + h * 64 + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8*)(src + ks * 16);
    }

    // ---- LDS-DMA: wave w moves rows 16 w .. 16 w + 15 of both operand tiles (two 1-KiB instructions each); ONE pointer per
    // operand and lane, the second instruction's rows r + 8 flip bit 2 of the chunk swizzle (+- 64 bytes)
    const int rsub = lane >> 3, pos = lane & 7;
    const int r0 = wave * 16 + rsub;
    const int sl0 = pos ^ ((r0 >> 1) & 7);
    const char* kp = (const char*)(QK + (row0 + r0) * ldqk + k_col_off + h * 64 + sl0 * 8);
    const char* vp = (const char*)(Vt + (int64_t)(h * 64 + r0) * ldvt + row0 + sl0 * 8);
    const int64_t flip = (sl0 & 4) ? -64 : 64;
    const int64_t k8 = (int64_t)8 * ldqk * 2 + flip, v8 = (int64_t)8 * ldvt * 2 + flip;
    const int64_t kstep = (int64_t)64 * ldqk * 2;
    auto issue = [&](int stage) {
        char* sK = smem + stage * A2_STAGE + wave * 2048;
        glds16(kp, sK);
        glds16(kp + k8, sK + 1024);
        glds16(vp, sK + 8192);
        glds16(vp + v8, sK + 8192 + 1024);
        kp += kstep;
        vp += 128;
    };

    // ---- LDS read addresses of this lane: A-operand fragment (row li of a 32-row block, 16-byte chunk 2 ks + hi)
    const int keyi = 16 * (li >> 4) + 8 * ((li >> 2) & 1) + 4 * ((li >> 3) & 1) + (li & 3);
    const int kb = keyi * 128 + ((hi ^ ((keyi >> 1) & 7)) << 4);
    const int vb = 8192 + li * 128 + ((hi ^ ((li >> 1) & 7)) << 4);
    int kaddr[4], vaddr[4];   // byte offsets into smem for the CURRENT stage of each operand (advanced once per tile)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kaddr[i] = kb ^ (i << 5);
        vaddr[i] = vb ^ (i << 5);
    }
    auto ldK = [&](bf16x8(&f)[4], int hb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = *(const bf16x8*)(smem + kaddr[ks] + hb * 4096);
    };
    auto ldV = [&](bf16x8(&f)[4], int hb) {   // f[2 ks + db]: key step 2 hb + ks of the tile, d block db
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db) f[2 * ks + db] = *(const bf16x8*)(smem + vaddr[2 * hb + ks] + db * 4096);
    };
    auto advance = [&](int(&a)[4], int stage_to) {   // stage_to: the stage the addresses move to
        const int d = stage_to == 0 ? -(A2_NSTAGE - 1) * A2_STAGE : A2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += d;
    };

    f32x16 S0, S1, negm, o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float ls0 = 0.f, ls1 = 0.f;     // this lane's share (its 16 of every 32 keys) of the row sum, two chains
    union { uint32_t u[4]; bf16x8 v; } pb[2];   // P^T B operands of the two key steps of a half tile
    bf16x8 fa[4], fb[4];

    // exp / row sum / pack of key step ks (registers 8 ks .. 8 ks + 7) of a score block
    auto softmax_step = [&](const f32x16& X, int ks) {
        float p[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(X[8 * ks + e]);
        ls0 += (p[0] + p[1]) + (p[2] + p[3]);   // built with -fno-slp-vectorize: packed f32 adds beside MFMAs cost more than plain ones
        ls1 += (p[4] + p[5]) + (p[6] + p[7]);
#pragma unroll
        for (int e = 0; e < 4; ++e) pb[ks].u[e] = pack2bf(p[2 * e], p[2 * e + 1]);
        asm volatile("" : "+v"(ls0), "+v"(ls1));   // the sums are complete inside this segment
    };
    auto qk = [&](f32x16& X, const bf16x8(&f)[4]) {
        X = WIW_MFMA32(f[0], qf[0], negm);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) X = WIW_MFMA32(f[ks], qf[ks], X);
    };
    auto pv = [&](const bf16x8(&f)[4], int ks) {
        o[0] = WIW_MFMA32(f[2 * ks + 0], pb[ks].v, o[0]);
        o[1] = WIW_MFMA32(f[2 * ks + 1], pb[ks].v, o[1]);
    };
    // raise the reference where a score of block X exceeds it by more than 2^THR (wave-uniform branch; rare after the first tiles)
    auto decide = [&](f32x16& X, float mx) {
        if (__builtin_amdgcn_ballot_w64(mx > A2_THR) != 0) {
            const float delta = max2r(partner_max(mx), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            const float nm = negm[0] - delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                X[r] -= delta;
                negm[r] = nm;
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            ls0 *= alpha;
            ls1 *= alpha;
        }
    };

    const int nkt = S / 64;
    // ---- prologue: tiles 0 and 1 in flight, tile 0's first half decides the initial reference
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nkt > 2) issue(2);
    ldK(fa, 0);
    ldK(fb, 1);
    {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        S0 = WIW_MFMA32(fa[0], qf[0], z);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) S0 = WIW_MFMA32(fa[ks], qf[ks], S0);
        const float m0 = partner_max(max16(S0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S0[r] -= m0;
            negm[r] = -m0;
        }
    }
    // segment C of tile 0
    qk(S1, fb);
    softmax_step(S0, 0);
    ldV(fa, 0);
    A32_INTERLEAVE(WIW_A32_VALU_A)
    pin4(fa);
    __builtin_amdgcn_sched_barrier(0);
    // segment D of tile 0 (tile 1 is already visible)
    advance(kaddr, 1);
    pv(fa, 0);
    softmax_step(S0, 1);
    pv(fa, 1);
    float mx1 = max16(S1);
    ldK(fb, 0);   // K(1, h0)
    A32_INTERLEAVE(WIW_A32_VALU_B)
    pin4(fb);
    __builtin_amdgcn_sched_barrier(0);
    decide(S1, mx1);

    int stage = 1;   // stage of tile j
    for (int j = 1; j < nkt; ++j) {
        // ---- A: S0 = K(j,h0).Q - m  |  S1 key step 0  |  reads V(j-1,h1) (vaddr still points at tile j-1)
        qk(S0, fb);
        softmax_step(S1, 0);
        ldV(fa, 1);
        A32_INTERLEAVE(WIW_A32_VALU_A)
        pin4(fa);
        __builtin_amdgcn_sched_barrier(0);
        // ---- B: O += V(j-1,h1).P1  |  S1 key step 1, max of S0  |  reads K(j,h1)
        pv(fa, 0);
        softmax_step(S1, 1);
        pv(fa, 1);
        const float mx0 = max16(S0);
        ldK(fb, 1);
        A32_INTERLEAVE(WIW_A32_VALU_B)
        pin4(fb);
        __builtin_amdgcn_sched_barrier(0);
        decide(S0, mx0);
        // ---- C: S1 = K(j,h1).Q - m  |  S0 key step 0  |  reads V(j,h0)
        advance(vaddr, stage);
        qk(S1, fb);
        softmax_step(S0, 0);
        ldV(fa, 0);
        A32_INTERLEAVE(WIW_A32_VALU_A)
        pin4(fa);
        __builtin_amdgcn_sched_barrier(0);
        // ---- tile j+1 visible to everybody; the stage of tile j-1 is free for tile j+2
        const int nstage = stage == A2_NSTAGE - 1 ? 0 : stage + 1;
        if (j + 1 < nkt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (j + 2 < nkt) issue(nstage == A2_NSTAGE - 1 ? 0 : nstage + 1);
        }
        // ---- D: O += V(j,h0).P0  |  S0 key step 1, max of S1  |  reads K(j+1,h0)
        advance(kaddr, nstage);
        pv(fa, 0);
        softmax_step(S0, 1);
        pv(fa, 1);
        mx1 = max16(S1);
        ldK(fb, 0);   // past the last tile: a stale stage, never used
        A32_INTERLEAVE(WIW_A32_VALU_B)
        pin4(fb);
        __builtin_amdgcn_sched_barrier(0);
        decide(S1, mx1);
        stage = nstage;
    }
    // ---- epilogue: the second half of the last tile (vaddr points at it: stage `stage` was advanced past... see below)
    // after the loop `stage` is the stage of tile nkt (one past the last); vaddr still points at tile nkt-1
    softmax_step(S1, 0);
    softmax_step(S1, 1);
    ldV(fa, 1);
    pv(fa, 0);
    pv(fa, 1);

    // ---- normalise and store.  Lane (q, hi) holds O[q][32 db + 8 i + 4 hi + (0..3)] in registers 4 i .. 4 i + 3 of o[db]; a
    // v_permlane32_swap per packed word gives the lower lane d = 16 j .. + 7 and the upper lane d = 16 j + 8 .. + 15: 16-byte stores
    const float l = xor32_sum(ls0 + ls1);
    const float inv = 1.0f / l;
    uint16_t* dst = O + (row0 + qrow) * ldo + h * 64 + hi * 8;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int re = 8 * jp, ro = 8 * jp + 4;   // registers of the even / odd 4-row group
            const uint32_t e0 = pack2bf(o[db][re] * inv, o[db][re + 1] * inv), e1 = pack2bf(o[db][re + 2] * inv, o[db][re + 3] * inv);
            const uint32_t o0 = pack2bf(o[db][ro] * inv, o[db][ro + 1] * inv), o1 = pack2bf(o[db][ro + 2] * inv, o[db][ro + 3] * inv);
            const auto s0 = __builtin_amdgcn_permlane32_swap(e0, o0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(e1, o1, false, false);
            *(uint4*)(dst + 32 * db + 16 * jp) = uint4{s0[0], s1[0], s0[1], s1[1]};
        }
}

}  // namespace

// wiw_attn_spatial_bf16 of attention.hip with the scale already applied (exp2 domain); the general fallback
int wiw_attn_spatial_prescaled_fallback(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt, void* O,
                                        int ldo, int frames, int S, int heads, const void* zeros);

extern "C" int wiw_attn_spatial_ps_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt,
                                        void* O, int ldo, int frames, int S, int heads, const void* zeros) {
    WIW_REQUIRE(QK && Vt && O && zeros, "attn_spatial_ps: null pointer");
    WIW_REQUIRE(frames > 0 && S > 0 && heads > 0, "attn_spatial_ps: bad sizes");
    WIW_REQUIRE(S % 8 == 0, "attn_spatial_ps: S (= h*w of the level) must be a multiple of 8");
    WIW_REQUIRE(ldqk % 8 == 0 && k_col_off % 8 == 0 && ldvt % 8 == 0 && ldo % 8 == 0, "attn_spatial_ps: misaligned strides");
    static const bool force_old = getenv("WIW_ATTN_V1") != nullptr;   // A/B knob: the first form on the pre-scaled operands
    if (S % 128 != 0 || S < 256 || force_old)
        return wiw_attn_spatial_prescaled_fallback(stream, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads, zeros);
    const int q_tiles = S / 128;
    const int64_t nb = (int64_t)q_tiles * heads * frames;
    WIW_REQUIRE(nb < (1ll << 31), "attn_spatial_ps: grid too large");
    hipLaunchKernelGGL(attn_spatial32_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)QK, ldqk,
                       k_col_off, (const uint16_t*)Vt, ldvt, (uint16_t*)O, ldo, S, heads, q_tiles);
    return wiw_check_launch("wiw_attn_spatial_ps_bf16");
}
