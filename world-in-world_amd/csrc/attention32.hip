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

#ifndef WIW_A32_FORM
#define WIW_A32_FORM 3    // 3: three-deep pipeline, Q.K and P.V MFMAs alternate; 2: hand-placed gaps, one segment per operand; 1: the first correct form
#endif
#ifndef WIW_A32_LSUM_MFMA
#define WIW_A32_LSUM_MFMA 1   // form 2: row sums by a 16x16x32 MFMA on the P registers (1) or by v_add_f32 (0)
#endif
#ifndef WIW_A32_ABLATE
#define WIW_A32_ABLATE 0   // timing experiments only (results wrong): 1 no softmax VALU (exp / pack / max / decision), 2 no K/V
#endif                     // DMA + no tile barrier, 4 no LDS reads, 8 Q.K MFMAs independent (C = 0 each), 16 no P.V MFMAs
#ifndef WIW_A32_WAVES
#define WIW_A32_WAVES 3   // waves per SIMD the register budget is cut for (3: 168 VGPRs — form 3 fits without a spill; 2: 256)
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

// v_max3_f32 / v_max_f32 through the BUILTIN (the file is built with -fno-honor-nans, so no canonicalising self-max is added).
// NOT through an asm statement: these read MFMA results, and hipcc pads the MFMA -> VALU read hazard (up to 12 wait states
// behind an 8-pass MFMA) only for instructions it models — an asm v_max3 issued too early reads a partly written
// accumulator.  Round 5 found exactly that: the maximum only steers the raise decision, so results stayed inside the
// tolerance but differed by an ulp from launch to launch (tools/attn32_probe.py REPEATS).
WIW_DEV float max3r(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
WIW_DEV float max2r(float a, float b) { return __builtin_fmaxf(a, b); }
WIW_DEV float max16(const f32x16& x) {   // 8 instructions, depth 3
    const float a = max3r(x[0], x[1], x[2]), b = max3r(x[3], x[4], x[5]), c = max3r(x[6], x[7], x[8]);
    const float d = max3r(x[9], x[10], x[11]), e = max3r(x[12], x[13], x[14]);
    return max2r(max3r(a, b, x[15]), max3r(c, d, e));
}
WIW_DEV float partner_max(float x) {   // max over the two lanes (hi = 0, 1) of a query
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max2r(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

#ifndef WIW_A32_DBG
#define WIW_A32_DBG 0   // race hunting: 1 barrier at the top of every tile, 2 vmcnt(0) right after each DMA issue, 4 four ring stages
#endif
constexpr int A2_STAGE = 16384, A2_NSTAGE = (WIW_A32_DBG & 4) ? 4 : 3;
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

struct Ph0 { static constexpr int value = 0; };
struct Ph1 { static constexpr int value = 1; };

__global__ __launch_bounds__(256, WIW_A32_WAVES) void attn_spatial32_kernel(const uint16_t* __restrict__ QK, int ldqk, int k_col_off,
                                                                const uint16_t* __restrict__ Vt, int64_t ldvt,
                                                                uint16_t* __restrict__ O, int ldo, int S, int heads, int q_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[A2_NSTAGE * A2_STAGE + ((WIW_A32_DBG & 16) ? 16384 : 0) + ((WIW_A32_DBG & 32) ? 8192 : 0)];   // DBG 16: pad to 64 KB (2 blocks per CU); 32: to 56 KB
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
    auto issue = [&](int stage, bool adv = true) {   // adv (wave-uniform): step the source to the next tile afterwards
        char* sK = smem + stage * A2_STAGE + wave * 2048;
        glds16(kp, sK);
        glds16(kp + k8, sK + 1024);
        glds16(vp, sK + 8192);
        glds16(vp + v8, sK + 8192 + 1024);
        kp += adv ? kstep : 0;
        vp += adv ? 128 : 0;
        if (WIW_A32_DBG & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

#if WIW_A32_FORM >= 2
    f32x16 S0, S1, o[2];
#else
    f32x16 S0, S1, negm, o[2];
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float ls0 = 0.f, ls1 = 0.f;     // this lane's share (its 16 of every 32 keys) of the row sum, two chains
    union { uint32_t u[4]; bf16x8 v; } pb[2];   // P^T B operands of the two key steps of a half tile
    bf16x8 fa[4], fb[4];
    const int nkt = S / 64;

#if WIW_A32_FORM == 3
    // ================= three-deep pipeline: no two consecutive MFMAs on the same accumulator =================
    // Measured on form 2 (profiles/r11f_attn32_ablation.txt): the five DEPENDENT 32x32x16 MFMAs of a Q.K segment cost 23 % of
    // the kernel (a dependent MFMA issues ~56 cycles after its predecessor, an independent one after 32) and hide VALU work
    // that needs no hiding.  Here half k issues   I  P0 Q0 P1 Q1 P2 Q2 P3 Q3   — Q.K of block X_k (I = the -m initialisation)
    // alternating with P.V of block X_(k-2), accumulators Xq, o[0], Xq, o[1], ... — while the VALU exponentiates X_(k-1):
    // three blocks in flight (X_k produced, X_(k-1) -> P, P of X_(k-2) consumed).  <= 4 VALU issues per gap.
    //   * a K / V^T fragment register set is refilled right after the MFMA that read it, for the half after this one: every
    //     LDS read has 7+ MFMA gaps to land (counted lgkmcnt waits, never 0 in the steady state);
    //   * the raise decision for X_(k-1) sits after gap 2 (its maximum needs Q3 of the previous half).  At that point P.V of
    //     X_(k-2) is one MFMA in: the rare branch therefore also rescales the not yet consumed P operands — exactly, because
    //     the reference lives on the INTEGER grid (and on the 16-bit grid of the operand type, see initB): a raise is a power
    //     of two — and takes the raise off X_k, whose accumulation has started from the old -m (cdna guide T13 (a));
    //   * one barrier per tile, after gap 2 of the second half: tile j+1 becomes visible (its K is first read in gap 3), the
    //     stage of tile j-1 (last read: V^T fragments refilled during the first half) is handed to DMA(j+2).
#define A32_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
// A wait statement names ONLY registers whose reads it covers: "+v" on a register with a read still in flight would license
// hipcc to copy it (a v_mov of stale data; cdna guide 5.7: "forms (ii)/(iii) pin order, not register allocation").
#define A32_WAITF(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#define A32_WAIT3(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]))
#define A32_WAIT1(n, r) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(r))
#define A32_GAP                                                 \
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 16, 0);       \
    __builtin_amdgcn_sched_barrier(0);
    float p[8];
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};   // row sums on the matrix pipe: see form 2 below
    bf16x8 onesA, initA, initB;
    {
        const bool on = lane == 0 || lane == 32 || lane == 17 || lane == 49;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            onesA[e] = on ? (short)WIW_ONE16 : (short)0;
            initA[e] = 0;
            initB[e] = 0;
        }
        initA[0] = hi == 0 ? (short)WIW_ONE16 : (short)0;
    }
    float mref;   // the reference of this lane's query: an integer representable in the 16-bit operand type
    auto grid = [](float x) { return bf2f(f2bf(__builtin_rintf(x))); };
    union PB { uint32_t u[4]; bf16x8 v; };
    PB pbA[2], pbB[2];   // P^T B operands (two key steps) of the even / odd blocks
    // vaddr starts one stage BEHIND tile 0: the second half of a tile moves it to that tile's stage before its V^T refills
#pragma unroll
    for (int i = 0; i < 4; ++i) vaddr[i] += (A2_NSTAGE - 1) * A2_STAGE;

    auto half3 = [&](auto ph_tag, auto sync_tag, auto qk_tag, auto pv_tag, f32x16& Xq, f32x16& Xe, PB(&pbN)[2], PB(&pbO)[2], int j,
                     int& stage) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool SYNC = decltype(sync_tag)::value != 0, DO_QK = decltype(qk_tag)::value != 0, DO_PV = decltype(pv_tag)::value != 0;
        constexpr bool FULL = DO_QK && DO_PV;
        constexpr int VK = PH == 0 ? 2 : 0;        // V^T refill: V(j-1, h1) during the first half, V(j, h0) during the second
        constexpr int KOFF = PH == 0 ? 4096 : 0;   // K refill: K(j, h1) / K(j+1, h0)
        const int nstage = stage == A2_NSTAGE - 1 ? 0 : stage + 1;
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        // ---- gap 0: I
        if (FULL) {   // every fragment read but the two youngest (fa[3], fb[3] of the previous half) has landed
            A32_WAIT3(2, fa);
            A32_WAIT3(2, fb);
        } else {
            A32_WAITF(0, fa);
            A32_WAITF(0, fb);
        }
        if (DO_QK) Xq = WIW_MFMA32(initA, initB, zero);
        if (DO_PV) lacc = WIW_MFMA(onesA, pbO[1].v, lacc);
        if (PH == 1) advance(vaddr, stage);
        A32_GAP
        // ---- gap 1: P0
        if (DO_PV) o[0] = WIW_MFMA32(fa[0], pbO[0].v, o[0]);
        A32_DSR(fa[0], vaddr[VK], 0);
        const float ma = max3r(Xe[0], Xe[1], Xe[2]), mb = max3r(Xe[3], Xe[4], Xe[5]), mc = max3r(Xe[6], Xe[7], Xe[8]);
        const float md = max3r(Xe[9], Xe[10], Xe[11]);
        A32_GAP
        // ---- gap 2: Q0
        if (DO_QK) Xq = WIW_MFMA32(fb[0], qf[0], Xq);
        const float me = max3r(Xe[12], Xe[13], Xe[14]);
        const float mx = max2r(max3r(ma, mb, Xe[15]), max3r(mc, md, me));
        A32_GAP
        if (__builtin_amdgcn_ballot_w64(mx > A2_THR) != 0) {   // wave-uniform, rare after the first tiles
            const float mq = partner_max(mx);
            const float m_new = grid(mref + (mq > A2_THR ? mq : 0.f));   // unchanged where mq <= THR
            const float delta = m_new - mref;                              // an integer: alpha is a power of two
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Xe[r] -= delta;
                if (DO_QK) Xq[r] -= delta;
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            if (DO_PV) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const wiw_f32x2 v = unpack2(pbO[ks].u[e]);
                        pbO[ks].u[e] = pack2bf(v.x * alpha, v.y * alpha);
                    }
            }
            lacc[0] *= alpha;                               // lanes 0..15: query n ...
            lacc[1] *= __shfl(alpha, (lane & 15) + 16);     // ... and query n + 16 (its factor lives in lane n + 16)
            mref = m_new;
            initB[0] = hi == 0 ? (short)f2bf(-m_new) : (short)0;
        }
        if (PH == 1 && SYNC) {   // NO branch here (hipcc sinks VALU work below a conditional block): past the end of the
            // sequence the DMA re-reads the last tile into the free stage
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(1)" ::: "memory");   // my DMA(j+1) pieces; my reads of tile j-1's stage
            __syncthreads();
            issue(nstage == A2_NSTAGE - 1 ? 0 : nstage + 1, j + 3 < nkt);
        }
        if (PH == 1) advance(kaddr, nstage);
        // ---- gap 3: P1
        if (DO_PV) o[1] = WIW_MFMA32(fa[1], pbO[0].v, o[1]);
        A32_DSR(fa[1], vaddr[VK], 4096);
        if (DO_QK) A32_DSR(fb[0], kaddr[0], KOFF);
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(Xe[e]);
        A32_GAP
        // ---- gap 4: Q1
        if (DO_QK) {
            Xq = WIW_MFMA32(fb[1], qf[1], Xq);
            A32_DSR(fb[1], kaddr[1], KOFF);
        }
#pragma unroll
        for (int e = 4; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(Xe[e]);
        A32_GAP
        // ---- gap 5: P2
        if (DO_PV) o[0] = WIW_MFMA32(fa[2], pbO[1].v, o[0]);
        A32_DSR(fa[2], vaddr[VK + 1], 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) pbN[0].u[e] = pack2bf(p[2 * e], p[2 * e + 1]);
        A32_GAP
        // ---- gap 6: Q2
        if (DO_QK) {
            Xq = WIW_MFMA32(fb[2], qf[2], Xq);
            A32_DSR(fb[2], kaddr[2], KOFF);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(Xe[8 + e]);
        A32_GAP
        // ---- gap 7: P3 (fa[3] was refilled in gap 7 of the previous half: 7 younger reads may be in flight)
        if (FULL) A32_WAIT1(7, fa[3]); else A32_WAITF(0, fa);
        if (DO_PV) o[1] = WIW_MFMA32(fa[3], pbO[1].v, o[1]);
        A32_DSR(fa[3], vaddr[VK + 1], 4096);
#pragma unroll
        for (int e = 4; e < 8; ++e) p[e] = __builtin_amdgcn_exp2f(Xe[8 + e]);
        A32_GAP
        // ---- gap 8: Q3
        if (DO_QK) {
            if (FULL) A32_WAIT1(7, fb[3]); else A32_WAITF(0, fb);
            Xq = WIW_MFMA32(fb[3], qf[3], Xq);
            A32_DSR(fb[3], kaddr[3], KOFF);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pbN[1].u[e] = pack2bf(p[2 * e], p[2 * e + 1]);
        lacc = WIW_MFMA(onesA, pbN[0].v, lacc);
        A32_GAP
        if (PH == 1) stage = nstage;
    };

    // ---- prologue: tiles 0 and 1 in flight; block X_0 (C = 0) decides the initial reference
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(2, nkt > 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) A32_DSR(fa[ks], kaddr[ks], 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) A32_DSR(fb[ks], kaddr[ks], 4096);
    A32_WAITF(4, fa);
    {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        S0 = WIW_MFMA32(fa[0], qf[0], z);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) S0 = WIW_MFMA32(fa[ks], qf[ks], S0);
        const float m0 = grid(partner_max(max16(S0)));
        mref = m0;
        initB[0] = hi == 0 ? (short)f2bf(-m0) : (short)0;
#pragma unroll
        for (int r = 0; r < 16; ++r) S0[r] -= m0;
    }
    int stage = 0;
    half3(Ph1{}, Ph0{}, Ph1{}, Ph0{}, S1, S0, pbA, pbB, 0, stage);                  // k = 1: Q.K of X_1, exp of X_0
    for (int j = 1; j < nkt; ++j) {
        half3(Ph0{}, Ph1{}, Ph1{}, Ph1{}, S0, S1, pbB, pbA, j, stage);              // k = 2 j
        half3(Ph1{}, Ph1{}, Ph1{}, Ph1{}, S1, S0, pbA, pbB, j, stage);              // k = 2 j + 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last two DMA issues (re-reads of the last tile) have landed
    half3(Ph0{}, Ph0{}, Ph0{}, Ph1{}, S0, S1, pbB, pbA, nkt, stage);                // k = 2 nkt: exp of the last block, P.V of the one before
    A32_WAITF(0, fa);
    lacc = WIW_MFMA(onesA, pbB[1].v, lacc);
    o[0] = WIW_MFMA32(fa[0], pbB[0].v, o[0]);
    o[1] = WIW_MFMA32(fa[1], pbB[0].v, o[1]);
    o[0] = WIW_MFMA32(fa[2], pbB[1].v, o[0]);
    o[1] = WIW_MFMA32(fa[3], pbB[1].v, o[1]);
#elif WIW_A32_FORM == 2
    // ================= hand-placed form: every MFMA gap is its own scheduling region =================
    // LDS reads are asm statements (they stay in their gap; hipcc's IR-level sinking otherwise moves a plain load next to
    // its user) with ONE s_waitcnt per fragment set at the end of the segment that issued it (cdna guide 5.7 form ii).
#define A32_DSR_(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define A32_WAIT4_(f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#if WIW_A32_ABLATE & 4
#define A32_DSR(dst, addr, off)
#define A32_WAIT4(f) asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]))
#else
#define A32_DSR(dst, addr, off) A32_DSR_(dst, addr, off)
#define A32_WAIT4(f) A32_WAIT4_(f)
#endif
#define A32_GAP                                                 \
    __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier(SG_VALU, 16, 0);       \
    __builtin_amdgcn_sched_barrier(0);
    float p[16];
#if WIW_A32_LSUM_MFMA
    // Row sums on the matrix pipe: one v_mfma_f32_16x16x32 per key step reads the SAME P registers as its B operand.  Under
    // the 16x16x32 operand map lane l supplies column l & 15, k group l >> 4 — i.e. column n collects query n (k groups 0, 2)
    // and query n + 16 (k groups 1, 3); an A operand whose row 0 is ones on k groups {0, 2} and row 1 ones on {1, 3} gives
    // D[0][n] = sum_k P[k][n], D[1][n] = sum_k P[k][n + 16]: registers 0 / 1 of lanes 0..15.  32 v_add per tile less, for 4
    // half-size MFMAs (the matrix pipe is the one with idle time; a SIMD retires about one instruction per 5.3 cycles
    // whatever its wave count, profiles/r11d_pmc_attn.csv, so the instruction count is the cost).
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 onesA;
    {
        const bool on = lane == 0 || lane == 32 || lane == 17 || lane == 49;
#pragma unroll
        for (int e = 0; e < 8; ++e) onesA[e] = on ? (short)WIW_ONE16 : (short)0;
    }
#define A32_LSUM(ks) lacc = WIW_MFMA(onesA, pb[ks].v, lacc)
#else
#define A32_LSUM(ks)
#endif
    // The accumulator of a Q.K segment starts at -m.  hipcc cannot take a 16-register C operand that differs from D without
    // copying it first (8 v_mov_b64 per block; behind the rescale branch it keeps TWO live copies of the block), and a fill
    // from a scalar costs the same.  The matrix pipe does it in ONE instruction: X = A1 . Bm with A1[i][0] = 1, Bm[0][n] = -m_n
    // (everything else 0, C = 0).  For that product to be exact m is kept on the 16-bit grid of the operand type (it only has
    // to bound the exponents), so a raise m -> m' has an exactly representable difference and exp2(m - m') rescales O and l
    // consistently.
    bf16x8 initA, initB;
#pragma unroll
    for (int e = 0; e < 8; ++e) { initA[e] = 0; initB[e] = 0; }
    initA[0] = hi == 0 ? (short)WIW_ONE16 : (short)0;
    float mref;   // the reference of this lane's query (a 16-bit-representable value)
    // One half iteration = a Q.K segment producing Xp (half tile PH of tile j) and a P.V segment consuming Xc (the half tile
    // before it), 9 MFMA gaps with <= 4 VALU issues each (a 32x32x16 MFMA hides four: tools/ubench/mfma_fill.hip T1):
    // exponentials and packs of Xc first (the P.V MFMAs need them), then the maximum of the fresh block.
#if WIW_A32_ABLATE & 1
#define A32_EXP(x) (x)
#define A32_PACK(dst, a, b)
#define A32_MAX3(a, b, c) (a)
#else
#define A32_EXP(x) __builtin_amdgcn_exp2f(x)
#define A32_PACK(dst, a, b) dst = pack2bf(a, b)
#define A32_MAX3(a, b, c) max3r(a, b, c)
#endif
#if WIW_A32_ABLATE & 8
#define A32_QKC(X) zero
#else
#define A32_QKC(X) X
#endif
#if WIW_A32_ABLATE & 16
#define A32_PV(acc, a, b) asm volatile("" : "+v"(acc) : "v"(a), "v"(b))
#else
#define A32_PV(acc, a, b) acc = WIW_MFMA32(a, b, acc)
#endif
    auto half = [&](auto ph_tag, auto sync_tag, f32x16& Xc, f32x16& Xp, int j, int& stage) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool SYNC = decltype(sync_tag)::value != 0;
        constexpr int VK = PH == 0 ? 2 : 0;              // key step (of the tile) of the consumed half tile's first V fragment
        constexpr int KOFF = PH == 0 ? 4096 : 0;         // next K fragments: K(j, h1) after phase 0, K(j+1, h0) after phase 1
        const int nstage = stage == A2_NSTAGE - 1 ? 0 : stage + 1;
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        // ---- Q.K segment
        Xp = WIW_MFMA32(initA, initB, zero);
        A32_DSR(fa[0], vaddr[VK], 0);
        A32_DSR(fa[1], vaddr[VK], 4096);
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = A32_EXP(Xc[e]);
        A32_GAP
        Xp = WIW_MFMA32(fb[0], qf[0], A32_QKC(Xp));
        A32_DSR(fa[2], vaddr[VK + 1], 0);
#pragma unroll
        for (int e = 4; e < 8; ++e) p[e] = A32_EXP(Xc[e]);
        A32_GAP
        Xp = WIW_MFMA32(fb[1], qf[1], A32_QKC(Xp));
        A32_DSR(fa[3], vaddr[VK + 1], 4096);
#pragma unroll
        for (int e = 0; e < 4; ++e) A32_PACK(pb[0].u[e], p[2 * e], p[2 * e + 1]);
        A32_GAP
        Xp = WIW_MFMA32(fb[2], qf[2], A32_QKC(Xp));
#pragma unroll
        for (int e = 8; e < 12; ++e) p[e] = A32_EXP(Xc[e]);
        A32_GAP
        Xp = WIW_MFMA32(fb[3], qf[3], A32_QKC(Xp));
#pragma unroll
        for (int e = 12; e < 16; ++e) p[e] = A32_EXP(Xc[e]);
        if (PH == 1) advance(kaddr, nstage);
        A32_GAP
        if (PH == 1 && SYNC && !(WIW_A32_ABLATE & 2)) {   // tile j+1 visible to everybody; the stage of tile j-1 is free for tile j+2.  NO branch here
            // (hipcc sinks the segment's VALU work below a conditional block): past the end of the sequence the DMA re-reads
            // the last tile into the free stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(nstage == A2_NSTAGE - 1 ? 0 : nstage + 1, j + 3 < nkt);
        }
        A32_WAIT4(fa);
        // ---- P.V segment
        A32_PV(o[0], fa[0], pb[0].v);
        A32_DSR(fb[0], kaddr[0], KOFF);
        A32_DSR(fb[1], kaddr[1], KOFF);
#pragma unroll
        for (int e = 0; e < 4; ++e) A32_PACK(pb[1].u[e], p[8 + 2 * e], p[8 + 2 * e + 1]);
#if WIW_A32_LSUM_MFMA
        A32_LSUM(0);
#else
        float t0 = p[0] + p[1], t1 = p[2] + p[3], t2 = p[4] + p[5], t3 = p[6] + p[7];
#endif
        A32_GAP
        A32_PV(o[1], fa[1], pb[0].v);
        A32_DSR(fb[2], kaddr[2], KOFF);
#if !WIW_A32_LSUM_MFMA
        t0 += t1; t2 += t3;
        ls0 += t0; ls1 += t2;
        float u0 = p[8] + p[9], u1 = p[10] + p[11], u2 = p[12] + p[13], u3 = p[14] + p[15];
#endif
        const float ma = A32_MAX3(Xp[0], Xp[1], Xp[2]), mb = A32_MAX3(Xp[3], Xp[4], Xp[5]), mc = A32_MAX3(Xp[6], Xp[7], Xp[8]);
        A32_GAP
        A32_PV(o[0], fa[2], pb[1].v);
        A32_DSR(fb[3], kaddr[3], KOFF);
        const float md = A32_MAX3(Xp[9], Xp[10], Xp[11]), me = A32_MAX3(Xp[12], Xp[13], Xp[14]);
        const float mf = A32_MAX3(ma, mb, Xp[15]), mg = A32_MAX3(mc, md, me);
#if WIW_A32_LSUM_MFMA
        A32_LSUM(1);
#else
        u0 += u1; u2 += u3;
#endif
        A32_GAP
        A32_PV(o[1], fa[3], pb[1].v);
        const float mx = (WIW_A32_ABLATE & 1) ? 0.f : max2r(mf, mg);
#if !WIW_A32_LSUM_MFMA
        ls0 += u0; ls1 += u2;
#endif
        if (PH == 0) advance(vaddr, stage);
        A32_GAP
        A32_WAIT4(fb);
        // Raise the reference where a score of Xp exceeds it by more than 2^THR (wave-uniform branch, rare after the first
        // tiles).  Everything at the old scale is here: O and l (all P.V MFMAs and row sums so far) and Xp.
        if (__builtin_amdgcn_ballot_w64(mx > A2_THR) != 0) {
            const float mq = partner_max(mx);
            const float m_new = bf2f(f2bf(mref + (mq > A2_THR ? mq : 0.f)));   // on the 16-bit grid; unchanged where mq <= THR
            const float delta = m_new - mref;                                    // exact
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Xp[r] -= delta;
                o[0][r] *= alpha;
                o[1][r] *= alpha;
            }
            mref = m_new;
            initB[0] = hi == 0 ? (short)f2bf(-m_new) : (short)0;
#if WIW_A32_LSUM_MFMA
            lacc[0] *= alpha;                                                     // lanes 0..15: query n ...
            lacc[1] *= __shfl(alpha, (lane & 15) + 16);                           // ... and query n + 16 (its factor lives in lane n + 16)
#else
            ls0 *= alpha;
            ls1 *= alpha;
#endif
        }
        if (PH == 1) stage = nstage;
    };

    // ---- prologue: tiles 0 and 1 in flight; the first half tile decides the initial reference
    issue(0);
    issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(2, nkt > 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) A32_DSR(fa[ks], kaddr[ks], 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) A32_DSR(fb[ks], kaddr[ks], 4096);
    A32_WAIT4(fa);
    {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        S0 = WIW_MFMA32(fa[0], qf[0], z);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) S0 = WIW_MFMA32(fa[ks], qf[ks], S0);
        const float m0 = bf2f(f2bf(partner_max(max16(S0))));   // on the 16-bit grid (see initB)
        mref = m0;
        initB[0] = hi == 0 ? (short)f2bf(-m0) : (short)0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S0[r] -= m0;
        }
    }
    A32_WAIT4(fb);
    int stage = 0;
    half(Ph1{}, Ph0{}, S0, S1, 0, stage);
    for (int j = 1; j < nkt; ++j) {
        half(Ph0{}, Ph1{}, S1, S0, j, stage);
        half(Ph1{}, Ph1{}, S0, S1, j, stage);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last two DMA issues (re-reads of the last tile) have landed
    // ---- epilogue: the second half of the last tile (vaddr points at the last tile)
    A32_DSR(fa[0], vaddr[2], 0);
    A32_DSR(fa[1], vaddr[2], 4096);
    A32_DSR(fa[2], vaddr[3], 0);
    A32_DSR(fa[3], vaddr[3], 4096);
#pragma unroll
    for (int e = 0; e < 16; ++e) p[e] = __builtin_amdgcn_exp2f(S1[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pb[0].u[e] = pack2bf(p[2 * e], p[2 * e + 1]);
        pb[1].u[e] = pack2bf(p[8 + 2 * e], p[8 + 2 * e + 1]);
    }
#if WIW_A32_LSUM_MFMA
    A32_LSUM(0);
    A32_LSUM(1);
#else
    ls0 += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[8] + p[9]) + (p[10] + p[11]));
    ls1 += ((p[4] + p[5]) + (p[6] + p[7])) + ((p[12] + p[13]) + (p[14] + p[15]));
#endif
    A32_WAIT4(fa);
    o[0] = WIW_MFMA32(fa[0], pb[0].v, o[0]);
    o[1] = WIW_MFMA32(fa[1], pb[0].v, o[1]);
    o[0] = WIW_MFMA32(fa[2], pb[1].v, o[0]);
    o[1] = WIW_MFMA32(fa[3], pb[1].v, o[1]);
#else
    // ================= first correct form (sched_group_barrier pattern per segment): kept as the A/B reference =================
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
        if (WIW_A32_DBG & 1) __syncthreads();
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
            if (WIW_A32_DBG & 8) {   // poison the stage of tile j-1 (what DMA(j+2) overwrites with three stages): who still reads it?
                const int pst = stage == 0 ? A2_NSTAGE - 1 : stage - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint4*)(smem + pst * A2_STAGE + (i * 256 + tid) * 16) = uint4{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};
            }
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

#endif

    // ---- normalise and store.  Lane (q, hi) holds O[q][32 db + 8 i + 4 hi + (0..3)] in registers 4 i .. 4 i + 3 of o[db]; a
    // v_permlane32_swap per packed word gives the lower lane d = 16 j .. + 7 and the upper lane d = 16 j + 8 .. + 15: 16-byte stores
#if (WIW_A32_LSUM_MFMA && WIW_A32_FORM == 2) || WIW_A32_FORM == 3
    (void)ls0; (void)ls1;
    const float l_lo = __shfl(lacc[0], li & 15), l_hi = __shfl(lacc[1], li & 15);   // query li: lane li & 15, register li >> 4
    const float l = li < 16 ? l_lo : l_hi;
#else
    const float l = xor32_sum(ls0 + ls1);
#endif
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
