// Spatial flash attention, second form (round 5): v_mfma_f32_32x32x16, head_dim 64, a three-deep software pipeline in which the
// Q.K MFMAs of one 32-key block alternate with the P.V MFMAs of the block two before it while the VALU exponentiates the
// block in between.  Measured at the served shapes (profiles/r12g_*): 1.05 PFLOP/s at S = 9216 x 5 heads, 1.00 at
// S = 2304 x 10 (first form, attention.hip: 0.93 - 0.96 / 0.86 - 0.93), bit-identical from launch to launch.
//
// What the measurements behind it say (tools/ubench/mfma_fill.hip -> profiles/r11a_mfma_fill_ubench.txt, r11d_pmc_attn_*.csv,
// r11e / r11f_attn32_*.txt):
//   * one wave hides FOUR independent VALU issues behind a 32x32x16 MFMA (a fifth costs +6 cycles), ONE behind a 16x16x32;
//   * a wave that issues its MFMAs as a burst and its softmax as a block keeps the matrix pipe 0.59-0.63 busy even with four
//     such waves per SIMD, the same instructions in the order { MFMA ; 4-8 VALU } 0.71-0.72: the order is written by hand
//     here, every MFMA gap its own scheduling region (sched_barrier), LDS reads as asm statements with counted waits;
//   * a SIMD retires roughly one instruction per 5 cycles whatever its wave count, so the instruction count is the cost:
//     the scale and the reference are folded into the matrix pipe (below), row sums are MFMAs — 76 VALU issues per 64-key
//     tile and wave against 118 in the first form, for 22 MFMAs against 36 (+ 16 LDS reads, one barrier);
//   * d = 64 is the hard case: per MFMA cycle the softmax has twice the work of a d = 128 head.
//
// Formulation (transposed, as the first form — probabilities never leave the registers, no operand is transposed in LDS):
//     S'^T = K . Q^T - m      A = K rows (32 keys x 16 d per MFMA), B = Q^T;   the accumulator starts at -m (below)
//     O^T += V^T . P^T        A = V^T rows (32 d x 16 keys),        B = P^T straight from the S'^T accumulator registers
// * Q arrives PRE-SCALED by log2(e) / sqrt(64): the factor is folded into the to_q rows of the q|k|v projection weights when
//   they are packed (unet.py ATTN_PRESCALE), before their one rounding to 16 bits.  The accumulator then holds the base-2
//   exponent itself: p = v_exp_f32(s') is the only instruction per score besides its share of the maximum and the pack.
// * The accumulator of a block starts at -m through the matrix pipe as well: X = A1 . Bm with A1[i][0] = 1, Bm[0][n] = -m_n
//   (everything else 0, C = 0) — ONE instruction.  hipcc cannot hand a 16-register C operand that differs from D to the MFMA
//   builtin without copying it (8 v_mov_b64 per block; behind the raise branch it kept TWO live copies, 32 registers).  For
//   the product to be exact m lives on the INTEGER grid (and is representable in the 16-bit operand type): it only has to
//   bound the exponents, and a raise m -> m' is then a power of two, exact on O, l and the packed P operands.
// * C/D layout of 32x32x16: lane (n = lane & 31, hi = lane >> 5), register r <-> row (r & 3) + 8 (r >> 2) + 4 hi.  The K row
//   that feeds MFMA row i is free: key(i) = 16 (i >> 4) + 8 ((i >> 2) & 1) + 4 ((i >> 3) & 1) + (i & 3) makes registers
//   8 ks .. 8 ks + 7 of lane (q, hi) the EIGHT CONSECUTIVE keys 16 ks + 8 hi .. + 7: exactly the B operand (k = 8 hi + e) of
//   the P.V MFMA of key step ks — no cross-lane traffic (the usual layout needs 8 v_permlane32_swap per 32 scores), and
//   the V^T A operand is one aligned 16-byte LDS read.
// * Row sums on the matrix pipe: one v_mfma_f32_16x16x32 per key step reads the SAME P registers as its B operand.  Under the
//   16x16x32 operand map lane l supplies column l & 15, k group l >> 4 — column n collects query n (k groups 0, 2) and query
//   n + 16 (k groups 1, 3); an A operand whose row 0 is ones on k groups {0, 2} and row 1 ones on {1, 3} gives
//   D[0][n] = sum_k P[k][n], D[1][n] = sum_k P[k][n + 16]: registers 0 / 1 of lanes 0..15.  32 v_add per tile less.
// * Lazy reference: m is the rounded row maximum of the first 32 keys and is raised (everything at the old scale rescaled
//   exactly once, cdna guide T13) only when some query of the wave exceeds it by more than 2^8; the test costs 8 v_max3 and one
//   compare per 16 scores, the cross-lane part runs in the rare branch only.
// * K / V^T tiles arrive by LDS-DMA into a ring of THREE 16-KiB stages, one barrier per tile.  LDS images: K tile
//   [64 keys][128 B], V^T tile [64 d][128 B]; the 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 7): every 16-lane group
//   of a ds_read_b128 (rows distinct mod 16, same chunk) covers all 64 banks once (SQ_LDS_BANK_CONFLICT = 0; first form: a third
//   of its LDS cycles).
//
// A hazard this file met (and attention.hip had, latently): v_max3_f32 written as an asm statement on MFMA results.  hipcc
// pads the MFMA -> VALU read hazard (up to 12 wait states behind an 8-pass MFMA) only for instructions it models; the asm
// v_max3 sometimes read a partly written accumulator.  The maximum only steers the raise decision, so results stayed inside
// the tolerance — and differed by an ulp from launch to launch (profiles/r12c_attn32_nonrepeatable_hunt.txt: it looked like
// an LDS ring race for a while; 2 instead of 3 blocks per CU hid it).  Everything that touches an accumulator is a builtin now.
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

#ifndef WIW_A32_WAVES
#define WIW_A32_WAVES 3   // waves per SIMD the register budget is cut for (3: 168 VGPRs, no spill; 2: 256 — measured 2 % slower)
#endif

// v_max3_f32 / v_max_f32 through the BUILTIN (the file is built with -fno-honor-nans, so no canonicalising self-max is added).
// NOT through an asm statement: these read MFMA results, and hipcc pads the MFMA -> VALU read hazard (up to 12 wait states
// behind an 8-pass MFMA) only for instructions it models — an asm v_max3 issued too early reads a partly written
// accumulator.  Round 5 found exactly that: the maximum only steers the raise decision, so results stayed inside the
// tolerance but differed by an ulp from launch to launch (tools/attn32_probe.py REPEATS).
WIW_DEV float max3r(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
WIW_DEV float max2r(float a, float b) { return __builtin_fmaxf(a, b); }
WIW_DEV float max16(const f32x16& x) {   // 8 instructions, depth 3 (prologue only)
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

struct Ph0 { static constexpr int value = 0; };
struct Ph1 { static constexpr int value = 1; };

__global__ __launch_bounds__(256, WIW_A32_WAVES) void attn_spatial32_kernel(const uint16_t* __restrict__ QK, int ldqk, int k_col_off,
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
    auto issue = [&](int stage, bool adv = true) {   // adv (wave-uniform): step the source to the next tile afterwards
        char* sK = smem + stage * A2_STAGE + wave * 2048;
        glds16(kp, sK);
        glds16(kp + k8, sK + 1024);
        glds16(vp, sK + 8192);
        glds16(vp + v8, sK + 8192 + 1024);
        kp += adv ? kstep : 0;
        vp += adv ? 128 : 0;
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
    auto advance = [&](int(&a)[4], int stage_to) {   // stage_to: the stage the addresses move to
        const int d = stage_to == 0 ? -(A2_NSTAGE - 1) * A2_STAGE : A2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] += d;
    };

    f32x16 S0, S1, o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    bf16x8 fa[4], fb[4];   // V^T / K fragment sets (A operands)
    const int nkt = S / 64;

    // ================= three-deep pipeline: no two consecutive MFMAs on the same accumulator =================
    // Half k issues   I  P0 Q0 P1 Q1 P2 Q2 P3 Q3   — Q.K of block X_k (I = the -m initialisation)
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
    f32x4 lacc = f32x4{0.f, 0.f, 0.f, 0.f};   // row sums on the matrix pipe (header)
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

    // ---- normalise and store.  Lane (q, hi) holds O[q][32 db + 8 i + 4 hi + (0..3)] in registers 4 i .. 4 i + 3 of o[db]; a
    // v_permlane32_swap per packed word gives the lower lane d = 16 j .. + 7 and the upper lane d = 16 j + 8 .. + 15: 16-byte stores
    const float l_lo = __shfl(lacc[0], li & 15), l_hi = __shfl(lacc[1], li & 15);   // query li: lane li & 15, register li >> 4
    const float l = li < 16 ? l_lo : l_hi;
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
