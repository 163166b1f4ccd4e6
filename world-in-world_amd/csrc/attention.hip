// Attention kernels for gfx950, head_dim 64, bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 softmax.
//
// Both kernels use the "transposed" formulation so that the softmax probabilities never leave the
// registers and no operand needs a transpose in LDS:
//     S^T = K . Q^T        A = K rows (keys x d),      B = Q rows (queries x d)
//     O^T = V^T . P^T      A = V^T rows (d x keys),    B = P^T  (from the S^T accumulator itself)
// The MFMA C/D layout (lane l, reg r: row = 4*(l>>4) + r, col = l & 15) gives every lane 4 rows of one QUERY column
// (l & 15) per fragment; two such fragments are the 8-element B operand of the next MFMA.  Spatial kernel: the K rows
// are assigned to fragment rows so that those 8 values are 8 CONSECUTIVE keys (see the S^T loop), i.e. the natural K
// enumeration, and the V^T A operand is one 16-byte LDS read.  Temporal kernel (<= 16 keys): row = key, the K index of
// the second MFMA enumerated as key(q, e) = 16*(e>>2) + 4*q + (e&3) (q = l>>4).  Query statistics live in the lanes
// that own the query column — the O^T accumulator has the same column ownership, so rescaling needs no cross-lane traffic.
#include <stdlib.h>

#include "common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
WIW_DEV void glds16(const char* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

constexpr float LOG2E = 1.4426950408889634f;
#ifndef WIW_ATTN_MIN_WAVES
#define WIW_ATTN_MIN_WAVES 2   // launch-bounds hint (waves per SIMD); the kernel needs 126 VGPRs = 4 waves per SIMD on its own
#endif                         // (5: capped at 96 VGPRs, 27 spilled in the tile loop: 7.2 instead of 3.5 ms at S = 9216)
#ifndef WIW_ATTN_ABLATE
#define WIW_ATTN_ABLATE 0   // timing experiments only (results wrong): 1 no max / exp2 (P = bf16(S)), 2 no K/V DMA + no tile
#endif                      // barrier (stale stage 0), 4 no P.V MFMAs, 8 no Q.K MFMAs

// v_max_f32 / v_max3_f32 without the canonicalising self-max hipcc puts in front of fmaxf (IEEE NaN quieting): the file is built
// with -fno-honor-nans, under which the BUILTIN gives the bare instruction (the softmax loop is VALU-issue bound, every
// instruction of it counts; a NaN score still ends in a NaN row, its exp2 is NaN whatever the reference maximum is).
// Round 5: these were asm statements until the 32x32x16 kernel (attention32.hip) showed what that risks — they read MFMA
// results, and hipcc pads the MFMA -> VALU read hazard only for instructions it models.
WIW_DEV float max_raw(float a, float b) { return __builtin_fmaxf(a, b); }
WIW_DEV float max3_raw(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
WIW_DEV float xor16_max_raw(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
WIW_DEV float xor32_max_raw(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ---------------------------------------------------------------------------------------------
// Spatial flash attention.  Block = NWV waves = 32 * NWV queries of one (frame, head); KV tiles of 64 keys
// double-buffered in LDS by LDS-DMA (K tile 8 KiB [key][d], V^T tile 8 KiB [d][key]).
// ---------------------------------------------------------------------------------------------
constexpr int KB = 64;   // queries per block: 32 per wave, NWV = 4 or 8 waves
constexpr int KV_STAGE = 16384;   // K tile + V^T tile

// NWV = 4 in production; NWV = 8 (WIW_ATTN_8WAVES=1, S >= 2048) is the measured-slower A/B form, see the launcher.
// WLSE (the fine-tuning forward): also writes the row log-sum-exp in the log2 domain, lse[(frame * heads + h) * S + q] =
// log2 sum_k 2^(s_qk c), from fp32 sums of the UN-rounded probabilities — what wiw_attn_bwd_bf16 otherwise recomputes with a
// whole Q.K^T pass (train.hip, pass 1 of the dQ kernel).  32 more VALU adds per tile; the inference instantiation has none.
template <int NWV, bool WLSE>
__global__ __launch_bounds__(NWV * 64, WIW_ATTN_MIN_WAVES) void attn_spatial_kernel(const uint16_t* __restrict__ QK, int ldqk, int k_col_off,
                                                               const uint16_t* __restrict__ Vt, int64_t ldvt,
                                                               uint16_t* __restrict__ O, int ldo, int S, int heads,
                                                               int q_tiles, float scale_log2e, const char* zeros,
                                                               float* __restrict__ lse) {
    constexpr int QB = NWV * 32;
    constexpr int NI = 8 / NWV;        // 8-row DMA instructions per wave and operand tile (64 rows): 2 or 1
    __shared__ __attribute__((aligned(16))) char smem[2 * KV_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware remap: an XCD walks a contiguous range of (frame, head, q_tile) so K/V stay in its L2
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int qt = bid % q_tiles;
    const int fh = bid / q_tiles;
    const int h = fh % heads, n = fh / heads;
    const int64_t row0 = (int64_t)n * S;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- Q^T B-operands straight from global: lane (query fr, d chunk fq) -> 8 consecutive d
    bf16x8 qf[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        int qi = qt * QB + wave * 32 + f * 16 + fr;
        qi = qi < S ? qi : S - 1;
        const uint16_t* src = QK + (row0 + qi) * ldqk + h * 64 + fq * 8;
        qf[f][0] = *(const bf16x8*)(src);
        qf[f][1] = *(const bf16x8*)(src + 32);
    }

    // ---- LDS-DMA sources for this lane: per-lane source pointers advance by one KV tile per iteration (64 keys:
    // 64 rows of QK for K, 128 bytes of a V^T row); only the last tile needs the clamped / zero-filled form.
    const int rsub = lane >> 3, pos = lane & 7;
    const int nkt = (S + KB - 1) / KB;
    // ONE K and ONE V^T pointer per lane (round 4): instruction i > 0 of a tile reads rows 8 i further on — for K a uniform
    // byte offset (the chunk swizzle (r & 7) does not change), for V^T a uniform offset +- 64 bytes (its swizzle (r >> 1) & 7
    // flips bit 2).  With four 64-bit pointers hipcc kept one in scratch (126 VGPRs at the 4-waves-per-SIMD cap): two
    // scratch loads, a scratch store and — before the fourth DMA could be issued — an s_waitcnt vmcnt(0) that also waited
    // for the three DMA loads just issued, i.e. a full memory round trip per tile and wave.
    const int r0 = wave * NI * 8 + rsub;                    // row of the tile for instruction 0 (key for K, d for V^T)
    const char* kp0 = (const char*)(QK + (row0 + r0) * ldqk + k_col_off + h * 64 + (pos ^ (r0 & 7)) * 8);
    const char* vp0 = (const char*)(Vt + (int64_t)(h * 64 + r0) * ldvt + row0 + (pos ^ ((r0 >> 1) & 7)) * 8);
    const int64_t kstep = (int64_t)KB * ldqk * 2;
    const int64_t k8 = (int64_t)8 * ldqk * 2;               // 8 K rows
    // V^T rows 8 further on: the chunk position pos ^ sw becomes pos ^ sw ^ 4
    const int64_t v8 = (int64_t)8 * ldvt * 2 + ((((pos ^ ((r0 >> 1) & 7)) & 4) != 0) ? -64 : 64);
    auto kp_ = [&](int i) -> const char* { return i == 0 ? kp0 : kp0 + k8; };
    auto vp_ = [&](int i) -> const char* { return i == 0 ? vp0 : vp0 + v8; };
    static_assert(NI <= 2, "instruction i > 1 of a tile would flip the V^T swizzle back");
    auto issue = [&](int stage, int kt) {
        char* sK = smem + stage * KV_STAGE + wave * NI * 1024;
        char* sV = smem + stage * KV_STAGE + 8192 + wave * NI * 1024;
        if ((kt + 1) * KB <= S) {   // full tile (wave-uniform)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                glds16(kp_(i), sK + i * 1024);
                glds16(vp_(i), sV + i * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = (wave * NI + i) * 8 + rsub;
                // K rows past the end re-read the last key (masked to -inf below); V^T chunks past the end read zeros
                const int key = kt * KB + r;
                const char* ksrc = key < S ? kp_(i) : kp_(i) - (int64_t)(key - (S - 1)) * ldqk * 2;
                glds16(ksrc, sK + i * 1024);
                const int key0 = kt * KB + (pos ^ ((r >> 1) & 7)) * 8;
                glds16(key0 < S ? vp_(i) : zeros, sV + i * 1024);
            }
        }
#ifndef WIW_ATTN_STATIC_KV   // ablation build: every tile re-reads KV tile 0 (cache-hot operands)
        kp0 += kstep; vp0 += KB * 2;
#endif
    };

    f32x4 o[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int f = 0; f < 2; ++f) o[d][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY};
    float lsum[2] = {0.f, 0.f};   // WLSE: this lane's share of sum_k p (fp32, un-rounded)
    // Row sums ride on the matrix pipe: O^T gets a 65th "d" row whose V^T entries are all ones, i.e. one extra MFMA per
    // (key step, query frag) with a constant A operand.  Every register of l_acc[f] then holds sum_k P[k][query fr]
    // (of the bf16-rounded P the numerator uses), with no VALU adds in the loop and no cross-lane reduction at the end.
    f32x4 l_acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)WIW_ONE16;

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
#if WIW_ATTN_ABLATE & 2
        const char* sK = smem;
#else
        if (kt + 1 < nkt) issue((kt + 1) & 1, kt + 1);
        const char* sK = smem + (kt & 1) * KV_STAGE;
#endif
        const char* sV = sK + 8192;
        // ---- S^T = K . Q^T  (4 key frags x 2 query frags x 2 d steps)
        f32x4 s[4][2];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            // Which key sits in which row of a fragment is free.  Row i of fragment kf holds key
            //     32*(kf>>1) + 8*(i>>2) + 4*(kf&1) + (i&3),
            // so that the S^T accumulators of fragments 2ks, 2ks+1 of lane (query fr, fq) are the EIGHT CONSECUTIVE keys
            // 32*ks + 8*fq .. +7: the P^T B operand of k-step ks in the natural K enumeration, and its V^T A operand is
            // one aligned 16-byte LDS read (the round-2 row = key mapping needed two 8-byte pieces 32 B apart per
            // operand: 24 v_mov per tile to assemble them).
            const int krow = (kf >> 1) * 32 + (fr >> 2) * 8 + (kf & 1) * 4 + (fr & 3);
            const char* rowp = sK + krow * 128;
            const bf16x8 k0 = *(const bf16x8*)(rowp + (((0 + fq) ^ (krow & 7)) << 4));
            const bf16x8 k1 = *(const bf16x8*)(rowp + (((4 + fq) ^ (krow & 7)) << 4));
#pragma unroll
            for (int f = 0; f < 2; ++f) {
#if WIW_ATTN_ABLATE & 8
                s[kf][f] = f32x4{0.f, 1.f, 2.f, 3.f};
                asm volatile("" : "+v"(s[kf][f]) : "v"(k0), "v"(k1));
#else
                f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                z = WIW_MFMA(k0, qf[f][0], z);
                s[kf][f] = WIW_MFMA(k1, qf[f][1], z);
#endif
            }
        }
        // ---- online softmax over keys (rows of S^T); lane owns query column fr of each query frag.
        // Scores stay unscaled: p = exp2(s*c - m) with c = log2(e)/sqrt(d) folded into one FMA; the running max
        // m is kept in the scaled domain.  v_exp_f32 is used raw (arguments <= 0; denormal results flush to 0).
        if ((kt + 1) * KB > S) {   // wave-uniform: only the last, partial tile masks keys >= S
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kt * KB + (kf >> 1) * 32 + fq * 8 + (kf & 1) * 4 + r >= S) s[kf][f][r] = -INFINITY;
        }
        uint32_t pb[2][4][2];   // [query frag][key frag][2 dwords] = packed bf16x4 B operands of the K = 16 MFMAs
#if WIW_ATTN_ABLATE & 1
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                pb[f][kf][0] = pack2bf(s[kf][f][0], s[kf][f][1]);
                pb[f][kf][1] = pack2bf(s[kf][f][2], s[kf][f][3]);
            }
#else
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float ma = max3_raw(s[0][f][0], s[0][f][1], s[0][f][2]);   // two interleaved chains (dependent VALU: 8.5 cycles each)
            float mb = max3_raw(s[2][f][0], s[2][f][1], s[2][f][2]);
            ma = max3_raw(ma, s[0][f][3], s[1][f][0]);
            mb = max3_raw(mb, s[2][f][3], s[3][f][0]);
            ma = max3_raw(ma, s[1][f][1], s[1][f][2]);
            mb = max3_raw(mb, s[3][f][1], s[3][f][2]);
            float mx = max3_raw(ma, mb, s[1][f][3]);
            mx = max_raw(mx, s[3][f][3]);
            mx = xor16_max_raw(mx);
            mx = xor32_max_raw(mx);
            // Lazy rescale: the running reference m_run only has to bound the exponents, not equal the maximum; it is
            // raised (and l, O rescaled) when some query of the wave exceeds it by more than 2^8 — after the first
            // tiles that is rare, so the 32-register rescale and its exp2 leave the steady-state loop.  The result
            // sum(p v) / sum(p) does not depend on the reference.
            const float m_new = max_raw(m_run[f], mx * scale_log2e);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run[f] + 8.0f) != 0) {   // wave-uniform branch
                const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);   // first tile: exp2(-inf) = 0
                m_run[f] = m_new;
                if (WLSE) lsum[f] *= alpha;
                l_acc[f][0] *= alpha; l_acc[f][1] *= alpha; l_acc[f][2] *= alpha; l_acc[f][3] *= alpha;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    o[d][f][0] *= alpha; o[d][f][1] *= alpha; o[d][f][2] *= alpha; o[d][f][3] *= alpha;
                }
            }
            // p = exp2(s*c - m_ref): one PACKED fma per two scores; fma / exp2 / convert interleaved by the compiler
            const wiw_f32x2 c2 = {scale_log2e, scale_log2e}, nm2 = {-m_run[f], -m_run[f]};
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const wiw_f32x2 t01 = __builtin_elementwise_fma(wiw_f32x2{s[kf][f][0], s[kf][f][1]}, c2, nm2);
                const wiw_f32x2 t23 = __builtin_elementwise_fma(wiw_f32x2{s[kf][f][2], s[kf][f][3]}, c2, nm2);
                const float p0 = __builtin_amdgcn_exp2f(t01.x), p1 = __builtin_amdgcn_exp2f(t01.y);
                const float p2 = __builtin_amdgcn_exp2f(t23.x), p3 = __builtin_amdgcn_exp2f(t23.y);
                if (WLSE) lsum[f] += (p0 + p1) + (p2 + p3);
                pb[f][kf][0] = pack2bf(p0, p1);
                pb[f][kf][1] = pack2bf(p2, p3);
            }
        }
#endif
        // ---- O^T += V^T . P^T  (4 d frags x 2 query frags x 2 key steps); both operands in the natural K enumeration
        // thanks to the key permutation of the S^T fragments above
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int drow = d * 16 + fr;
            const char* rowp = sV + drow * 128;
            const int sw = (drow >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                union { uint32_t u[4]; bf16x8 v; } va;
                va.v = *(const bf16x8*)(rowp + (((ks * 4 + fq) ^ sw) << 4));   // keys 32*ks + 8*fq .. +7 of d row drow
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    union { uint32_t u[4]; bf16x8 v; } pv;
                    pv.u[0] = pb[f][2 * ks][0]; pv.u[1] = pb[f][2 * ks][1];
                    pv.u[2] = pb[f][2 * ks + 1][0]; pv.u[3] = pb[f][2 * ks + 1][1];
#if WIW_ATTN_ABLATE & 4
                    asm volatile("" : "+v"(o[d][f]) : "v"(va.v), "v"(pv.v));
#else
                    o[d][f] = WIW_MFMA(va.v, pv.v, o[d][f]);
                    if (d == 0) l_acc[f] = WIW_MFMA(ones, pv.v, l_acc[f]);
#endif
                }
            }
        }
#if !(WIW_ATTN_ABLATE & 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
    }
    // ---- normalise and store: lane holds O[query fr][d = dfrag*16 + 4*fq + r]
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float inv = 1.0f / l_acc[f][0];
        const int qi = qt * QB + wave * 32 + f * 16 + fr;
        if (WLSE) {   // the four fq lanes of a query column hold disjoint keys
            const float tot = xor32_sum(xor16_sum(lsum[f]));
            if (qi < S && fq == 0) lse[((int64_t)n * heads + h) * S + qi] = m_run[f] + __builtin_amdgcn_logf(tot);
        }
        if (qi < S) {
            uint16_t* dst = O + (row0 + qi) * ldo + h * 64 + fq * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint2 pk;
                pk.x = pack2bf(o[d][f][0] * inv, o[d][f][1] * inv);
                pk.y = pack2bf(o[d][f][2] * inv, o[d][f][3] * inv);
                *(uint2*)(dst + d * 16) = pk;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Temporal attention: one wave per (batch item, spatial site, head); the T <= 16 frames of a site
// are rows m = (b*T + t)*S + s of the QKV buffer (stride S*ld) — the permutes are address math.
// K and Q fragments come straight from global (16 B per lane); V is staged in a per-wave LDS tile
// and read back transposed (2-byte reads).  Keys 16..31 of the K=32 MFMA are zero padding.
// ---------------------------------------------------------------------------------------------
constexpr int TV_LD = 72;   // bf16 elements per staged V row (144 B: 16-B aligned, skewed banks)

__global__ __launch_bounds__(256) void attn_temporal_kernel(const uint16_t* __restrict__ QKV, int ldqkv,
                                                             uint16_t* __restrict__ O, int ldo, int T, int S, int heads,
                                                             int64_t total, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) uint16_t vst[4][16 * TV_LD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= total) return;     // wave-uniform; no block-level barrier below
    const int C = heads * 64;
    const int h = (int)(task % heads);
    const int64_t site = task / heads;           // b*S + s
    const int s = (int)(site % S);
    const int64_t b = site / S;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t frame_stride = (int64_t)S * ldqkv;
    const uint16_t* base = QKV + ((b * T) * S + s) * ldqkv + h * 64;

    // Q (B operand: query fr, d chunk) and K (A operand: key fr, d chunk)
    const int tq = fr < T ? fr : T - 1;
    const uint16_t* qsrc = base + tq * frame_stride + fq * 8;
    const bf16x8 q0 = *(const bf16x8*)(qsrc), q1 = *(const bf16x8*)(qsrc + 32);
    bf16x8 k0 = *(const bf16x8*)(qsrc + C), k1 = *(const bf16x8*)(qsrc + C + 32);
    if (fr >= T) {
        k0 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        k1 = k0;
    }
    // V rows -> LDS (row = key, 64 d); rows >= T are zero
    uint16_t* vs = vst[wave];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int key = i * 8 + (lane >> 3);
        uint4 v = uint4{0u, 0u, 0u, 0u};
        if (key < T) v = *(const uint4*)(base + key * frame_stride + 2 * C + (lane & 7) * 8);
        *(uint4*)(vs + key * TV_LD + (lane & 7) * 8) = v;
    }
    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
    st = WIW_MFMA(k0, q0, st);
    st = WIW_MFMA(k1, q1, st);
    // softmax over keys 4*fq + r for query fr
    float sv[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sv[r] = (fq * 4 + r) < T ? st[r] * scale_log2e : -INFINITY;
        mx = fmaxf(mx, sv[r]);
    }
    mx = xor16_max(mx);
    mx = xor32_max(mx);
    float p[4], l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p[r] = exp2f(sv[r] - mx); l += p[r]; }
    l = xor16_sum(l);
    l = xor32_sum(l);
    const float inv = 1.0f / l;
    union { uint32_t u[4]; bf16x8 v; } pv;
    pv.u[0] = pack2bf(p[0], p[1]); pv.u[1] = pack2bf(p[2], p[3]); pv.u[2] = 0u; pv.u[3] = 0u;
    // make the staged V visible to the whole wave (same wave wrote it; wait for the LDS writes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool store = fr < T;
    uint16_t* dst = O + ((b * T + tq) * S + s) * ldo + h * 64 + fq * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int dcol = d * 16 + fr;
        union { uint32_t u[4]; bf16x8 v; } va;
        const uint32_t e0 = vs[(fq * 4 + 0) * TV_LD + dcol], e1 = vs[(fq * 4 + 1) * TV_LD + dcol];
        const uint32_t e2 = vs[(fq * 4 + 2) * TV_LD + dcol], e3 = vs[(fq * 4 + 3) * TV_LD + dcol];
        va.u[0] = e0 | (e1 << 16); va.u[1] = e2 | (e3 << 16); va.u[2] = 0u; va.u[3] = 0u;
        f32x4 ot = f32x4{0.f, 0.f, 0.f, 0.f};
        ot = WIW_MFMA(va.v, pv.v, ot);
        if (store) {
            uint2 pk;
            pk.x = pack2bf(ot[0] * inv, ot[1] * inv);
            pk.y = pack2bf(ot[2] * inv, ot[3] * inv);
            *(uint2*)(dst + d * 16) = pk;
        }
    }
}

}  // namespace

static int attn_spatial_launch(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt, void* O, int ldo,
                               int frames, int S, int heads, float scale_log2e, const void* zeros, float* lse) {
    WIW_REQUIRE(QK && Vt && O && zeros, "attn_spatial: null pointer");
    WIW_REQUIRE(frames > 0 && S > 0 && heads > 0, "attn_spatial: bad sizes");
    WIW_REQUIRE(S % 8 == 0, "attn_spatial: S (= h*w of the level) must be a multiple of 8");
    WIW_REQUIRE(ldqk % 8 == 0 && k_col_off % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0, "attn_spatial: misaligned strides");
    // 8 waves (256 queries) per staged KV tile halve the LDS-DMA instructions per flop but measured 9-10 % SLOWER at
    // S = 9216 / 2304 (3.80 vs 3.45 ms, profiles/r06q_attn_probe.txt: twice the waves behind every tile barrier); A/B knob
    static const bool want8 = getenv("WIW_ATTN_8WAVES") != nullptr;
    const bool big = S >= 2048 && want8 && lse == nullptr;
    const int qb = big ? 256 : 128;
    const int q_tiles = (S + qb - 1) / qb;
    const int64_t nb = (int64_t)q_tiles * heads * frames;
    WIW_REQUIRE(nb < (1ll << 31), "attn_spatial: grid too large");
#define WIW_ATTN_LAUNCH(NW, WL)                                                                                                    \
    hipLaunchKernelGGL((attn_spatial_kernel<NW, WL>), dim3((unsigned)nb), dim3(NW * 64), 0, (hipStream_t)stream, (const uint16_t*)QK, \
                       ldqk, k_col_off, (const uint16_t*)Vt, ldvt, (uint16_t*)O, ldo, S, heads, q_tiles, scale_log2e,              \
                       (const char*)zeros, lse)
    if (lse) WIW_ATTN_LAUNCH(4, true);
    else if (big) WIW_ATTN_LAUNCH(8, false);
    else WIW_ATTN_LAUNCH(4, false);
#undef WIW_ATTN_LAUNCH
    return wiw_check_launch("wiw_attn_spatial_bf16");
}

extern "C" int wiw_attn_spatial_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt,
                                     int64_t ldvt, void* O, int ldo, int frames, int S, int heads, float scale,
                                     const void* zeros) {
    return attn_spatial_launch(stream, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads, scale * LOG2E, zeros, nullptr);
}

// The first form on operands whose Q is pre-scaled by log2(e) / sqrt(d) (wiw_attn_spatial_ps_bf16, attention32.hip): the
// sequences the 32x32x16 kernel does not take (S % 128 != 0 or S < 256: the 18x32 / 9x16 levels)
int wiw_attn_spatial_prescaled_fallback(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt, void* O,
                                        int ldo, int frames, int S, int heads, const void* zeros) {
    return attn_spatial_launch(stream, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads, 1.0f, zeros, nullptr);
}

extern "C" int wiw_attn_spatial_lse_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt,
                                         int64_t ldvt, void* O, int ldo, int frames, int S, int heads, float scale,
                                         const void* zeros, float* lse) {
    WIW_REQUIRE(lse, "attn_spatial_lse: null lse pointer");
    return attn_spatial_launch(stream, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads, scale * LOG2E, zeros, lse);
}

extern "C" int wiw_attn_temporal_bf16(void* stream, const void* QKV, int ldqkv, void* O, int ldo, int batch, int T,
                                      int S, int heads, float scale) {
    WIW_REQUIRE(QKV && O, "attn_temporal: null pointer");
    WIW_REQUIRE(batch > 0 && S > 0 && heads > 0, "attn_temporal: bad sizes");
    WIW_REQUIRE(T >= 1 && T <= 16, "attn_temporal: 1 <= T <= 16");
    WIW_REQUIRE(ldqkv % 8 == 0 && ldqkv >= 3 * heads * 64 && ldo % 4 == 0, "attn_temporal: misaligned strides");
    const int64_t total = (int64_t)batch * S * heads;
    const int64_t nb = (total + 3) / 4;
    WIW_REQUIRE(nb < (1ll << 31), "attn_temporal: grid too large");
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)QKV,
                       ldqkv, (uint16_t*)O, ldo, T, S, heads, total, scale * LOG2E);
    return wiw_check_launch("wiw_attn_temporal_bf16");
}
