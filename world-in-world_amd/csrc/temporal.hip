// Fused temporal self-attention block for gfx950 (MI355X): LayerNorm -> Q/K/V projection -> 14x14 softmax(QK^T)V,
// one kernel, the 3C-wide QKV tensor never reaches HBM (TemporalBasicTransformerBlock.norm1 + attn1 up to its
// out-projection, reference dp/models/attention.py:735-737 with attention_processor.py:2358-2391).
//
//   O[(b,t,s)][h*64 + d] = sum_t' softmax_t'( q(b,t,s,h) . k(b,t',s,h) * scale ) v(b,t',s,h)[d]
//   (q | k | v)(row) = LayerNorm(X[row]; gamma, beta, eps) . W{q,k,v}^T            rows m = (b*T + t)*S + s
//
// Work item = (tile of 16 consecutive sites of one batch item, head).  The item is a 256 x 192 x C GEMM whose rows
// are the 16 frame slots of 16 sites (only the T real frames are staged in LDS: 16*T rows) and whose columns are
// [q_h | k_h | v_h]; 8 waves, wave w owns sites 2w, 2w+1 = two 16-row MFMA fragments x all 192 columns (96
// accumulator VGPRs), so EVERYTHING the attention of a site needs lives in one wave's registers:
//   * the pipeline is the one of gemm.hip's 256x160 tile: persistent blocks, 3-stage LDS ring filled by
//     global_load_lds_dwordx4 (7 DMA instructions per wave per K tile, spread over the four slots), XOR-swizzled
//     128-byte rows, two wave groups one slot apart, counted vmcnt, next item's first two K tiles prefetched before
//     the epilogue;
//   * LayerNorm is FOLDED (exact algebra): the MFMAs run on the raw rows x and the weights W' = bf16(W * gamma);
//     q_n = rstd * (acc_n - mean * s_n) + t_n with s_n = sum_k W'_nk, t_n = sum_k W_nk beta_k.  The row statistics
//     are accumulated by the VALU from the very A fragments the MFMAs consume (each lane sees 8 of the 32 k-values of
//     its row per k-step), in the shadow of the matrix pipe — no LayerNorm pass, no statistics pass;
//   * no transposes: Q and K are accumulated with swapped operands (lane = row, 4 consecutive d), V un-swapped
//     (lane = d, 4 consecutive frames).  With the MFMA K index enumerated d(fq, e) = 16*(e>>2) + 4*fq + (e&3), two
//     Q (K) accumulator fragments ARE the 8-element B (A) operand of S^T = K.Q^T; S^T (lane = query, 4 consecutive
//     keys) IS the B operand of O^T = V^T.P^T and the V accumulator IS its A operand (keys 16..31 of the K = 32
//     MFMA are zero); O^T comes out lane = query row, 4 consecutive d;
//   * softmax over the <= 14 keys in registers (fp32, exp2 with the scale folded, v_permlane16/32_swap reductions);
//   * O goes through a per-wave LDS transpose and leaves as whole 128-byte rows (16-byte stores).
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "common.h"


#ifndef WIW_T_ABLATE
#define WIW_T_ABLATE 0    // resident form, timing experiments only: 1 = no epilogue, 2 = no A loads, 4 = no LDS reads of W, 8 = no row statistics
#endif
#ifndef WIW_T_STAGGER
#define WIW_T_STAGGER 1   // resident form: waves 4..7 start half an item late (0: all waves in lock-step, A/B)
#endif
#ifndef WIW_T_MFMA_STATS
#define WIW_T_MFMA_STATS 1   // 1: LayerNorm row statistics on the MATRIX pipe (round 4); 0: 16 v_dot2c per k-step (round 3, A/B)
#endif
// Row statistics on the matrix pipe: two extra MFMAs per site and k-step on operands the wave holds anyway,
//     S = ONES . A^T  ->  every register of lane (frow = m, *) accumulates  sum_k x[m][k]          (ONES: a constant fragment)
//     G = A . A^T     ->  lane (frow = m, fq = m >> 2), register m & 3 accumulates  sum_k x[m][k]^2  (the diagonal)
// exact fp32 sums of exact products, as v_dot2c gave — and no VALU instruction in the main loop: round 3's ablation priced
// the 16 v_dot2c per k-step at 45 of the resident form's 283 us (a VALU stream next to a streaming MFMA partner gets one
// issue per MFMA; the matrix pipe takes the four extra MFMAs per k-step for +17 % of its own time).
// (A first form took both moments from ONE MFMA per site by making fragment row 15 — a padding frame slot — a row of ones,
// with the loads of those lanes masked out through EXEC so that their registers kept the ones: hipcc re-shuffles such
// loop-carried "+v" registers with v_mov WHILE the asm loads are in flight — stale copies and loads landing in re-assigned
// registers, a memory fault on the GPU.  tools/experiments/temporal_ones_row.txt.)
#ifndef WIW_T_RES_S1_MFMA
#define WIW_T_RES_S1_MFMA 0   // resident form (256-register limit): 1 = also the plain row sums S on the matrix pipe (the ones
                              // fragment + its accumulators cost 12 registers: 32 bytes of scratch per lane, reloaded every K
                              // tile); 0 = S stays on v_dot2c — 8 VALU instructions per k-step instead of 16 — and only G, the
                              // sums of squares, moves to the matrix pipe
#endif
#ifndef WIW_T_EPI
#define WIW_T_EPI 1       // 1: LayerNorm fold / output scaling of the epilogue on packed fp32 pairs (0: scalar FMAs, round 2)
#endif

namespace {

constexpr int BK = 64;
constexpr int NF = 12;                            // 16-column fragments per wave: 4 q | 4 k | 4 v
constexpr int W_ROWS = NF * 16;                   // 192
constexpr int W_BYTES = W_ROWS * BK * 2;          // 24576
constexpr int NW = 8, STAGES = 3, D = 2;
constexpr int SCR_BYTES = 2048;                   // per-item fold vectors: s[192] | t[192] | pad (fp32), double-buffered
constexpr int STG_ROWB = 144;                     // staged O row: 64 bf16 + 16 B skew
constexpr float LOG2E = 1.4426950408889634f;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA as inline asm, M0 (the LDS address) saved / restored inside the statement.  NOT the builtin: hipcc knows that the
// builtin writes LDS and puts `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint — the epilogue's
// fold-vector reads waited ~1 000 cycles for the next item's prefetch that had just been issued (r06c trace).  Completion
// is counted by hand (wait_vmcnt), as it already was.  Source = wave-uniform base + 32-bit per-lane byte offset.
// (the LDS destination is passed as a 32-bit LDS address, LDSA(pointer into smem): a generic -> LDS pointer cast of a
// pointer the compiler cannot trace back to smem emits a null check that this hipcc cannot select on gfx950)
#define LDSA(ptr) (smem_lds + (uint32_t)((ptr) - smem))
WIW_DEV void glds16(const char* base, uint32_t ofs, uint32_t lds_addr) {
    const uint32_t lds = __builtin_amdgcn_readfirstlane(lds_addr);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(ofs), "s"(base), "s"(lds) : "memory");
}
template <int N>
WIW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
WIW_DEV wiw_f32x2 pk_fma(wiw_f32x2 a, wiw_f32x2 b, wiw_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
WIW_DEV void glds16p(const char* g, uint32_t lds_addr) {   // the same with a per-lane 64-bit source address
    const uint32_t lds = __builtin_amdgcn_readfirstlane(lds_addr);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
// wave-uniform runtime count -> immediate (counts of this kernel are <= 1 + 2*3 + 4 + 4 + 3 = 18; more is clamped = stricter)
WIW_DEV void wait_vmcnt_rt(int n) {
    switch (n < 18 ? n : 18) {
#define WIW_VM_CASE(k) case k: wait_vmcnt<k>(); break;
        WIW_VM_CASE(0) WIW_VM_CASE(1) WIW_VM_CASE(2) WIW_VM_CASE(3) WIW_VM_CASE(4) WIW_VM_CASE(5) WIW_VM_CASE(6)
        WIW_VM_CASE(7) WIW_VM_CASE(8) WIW_VM_CASE(9) WIW_VM_CASE(10) WIW_VM_CASE(11) WIW_VM_CASE(12) WIW_VM_CASE(13)
        WIW_VM_CASE(14) WIW_VM_CASE(15) WIW_VM_CASE(16) WIW_VM_CASE(17) WIW_VM_CASE(18)
#undef WIW_VM_CASE
        default: wait_vmcnt<0>(); break;
    }
}
WIW_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sink of the O stores of lanes without a valid row (the store count per wave stays a compile-time constant)
__device__ uint4 t_dump[512 * 64];

#ifdef WIW_T_TRACE   // debug build: block 0 of the resident form stamps s_memtime per wave for ONE steady-state item
__device__ long long g_rtrace[8][16];
constexpr int TR_ITEM = 3;
#endif

struct TemporalArgs {
    const uint16_t* X;      // bf16 [batch*T*S][C]
    const uint16_t* W;      // bf16 [heads*192][C] TILED (1-KiB blocks, see setup_loader); rows of head h: q_h | k_h | v_h, gamma folded
    const float* fold;      // fp32 [heads][512]     s[192] | t[192] | pad
    uint16_t* O;            // bf16 [batch*T*S][ldo]
    const char* zeros;
    int ldo, batch, T, S, C, heads;
    int tiles_per_batch;    // ceil(S / 16)
    int items;              // batch * tiles_per_batch * heads
    float eps, scale_log2e, inv_c;
};

// ---- the attention of a wave's two sites from its accumulators (per wave, no block barrier): LayerNorm fold, S = QK^T,
// softmax over the <= 14 keys, O = PV, O rows through the wave's LDS staging area to 128-byte rows in HBM.
// fs: the head's fold vectors s[192] | t[192] in LDS; stg: this wave's staging area (NST * 8 rows of STG_ROWB bytes).
template <int TP>
WIW_DEV void attention_epilogue(f32x4 (&acc)[2][NF], const float (&sum1)[2], const float (&sum2)[2], const float* fs, char* stg,
                                const TemporalArgs& p, int b_cur, int s0_cur, int h_cur, int wave, int lane) {
    constexpr int NST = (2 * TP + 7) / 8;
    // the lane's index arithmetic below is loop-invariant; hoisted out of the item loop it stays live through the main loop
    // (at the 256-register limit: a spill, reloaded behind `s_waitcnt vmcnt(0)` = behind the next item's A prefetch).
    // An opaque copy of the lane id keeps it inside the epilogue.
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
    const int rsub = lane >> 3;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        // LayerNorm statistics of row (site mi, frame frow): every lane of the row holds the totals
        const float s1 = xor32_sum(xor16_sum(sum1[mi])), s2 = xor32_sum(xor16_sum(sum2[mi]));
        const float mean = s1 * p.inv_c;
        float var = __builtin_fmaf(-mean, mean, s2 * p.inv_c);
        var = var < 0.f ? 0.f : var;
        const float rs = rsqrtf(var + p.eps);
        const float nrm = -rs * mean;
        // q, k: x = rs * acc + (t - rs*mean*s)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const float4 sv = *(const float4*)(fs + ni * 16 + fq * 4);
            const float4 tv = *(const float4*)(fs + W_ROWS + ni * 16 + fq * 4);
#if WIW_T_EPI == 1   // the same fp32 FMAs as packed pairs (v_pk_fma_f32: half the VALU instructions; no MFMA is in flight here)
            const wiw_f32x2 rs2 = {rs, rs}, nrm2 = {nrm, nrm};
            const wiw_f32x2 lo = pk_fma(wiw_f32x2{acc[mi][ni][0], acc[mi][ni][1]}, rs2, pk_fma(nrm2, wiw_f32x2{sv.x, sv.y}, wiw_f32x2{tv.x, tv.y}));
            const wiw_f32x2 hi = pk_fma(wiw_f32x2{acc[mi][ni][2], acc[mi][ni][3]}, rs2, pk_fma(nrm2, wiw_f32x2{sv.z, sv.w}, wiw_f32x2{tv.z, tv.w}));
            acc[mi][ni] = f32x4{lo.x, lo.y, hi.x, hi.y};
#else
            acc[mi][ni][0] = __builtin_fmaf(acc[mi][ni][0], rs, __builtin_fmaf(nrm, sv.x, tv.x));
            acc[mi][ni][1] = __builtin_fmaf(acc[mi][ni][1], rs, __builtin_fmaf(nrm, sv.y, tv.y));
            acc[mi][ni][2] = __builtin_fmaf(acc[mi][ni][2], rs, __builtin_fmaf(nrm, sv.z, tv.z));
            acc[mi][ni][3] = __builtin_fmaf(acc[mi][ni][3], rs, __builtin_fmaf(nrm, sv.w, tv.w));
#endif
        }
        // S^T = K . Q^T over d = 64 (two k-steps), lane: query frow, keys 4*fq + r
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            union { uint32_t u[4]; bf16x8 v; } qo, ko;
            qo.u[0] = pack2bf(acc[mi][2 * kk][0], acc[mi][2 * kk][1]);
            qo.u[1] = pack2bf(acc[mi][2 * kk][2], acc[mi][2 * kk][3]);
            qo.u[2] = pack2bf(acc[mi][2 * kk + 1][0], acc[mi][2 * kk + 1][1]);
            qo.u[3] = pack2bf(acc[mi][2 * kk + 1][2], acc[mi][2 * kk + 1][3]);
            ko.u[0] = pack2bf(acc[mi][4 + 2 * kk][0], acc[mi][4 + 2 * kk][1]);
            ko.u[1] = pack2bf(acc[mi][4 + 2 * kk][2], acc[mi][4 + 2 * kk][3]);
            ko.u[2] = pack2bf(acc[mi][4 + 2 * kk + 1][0], acc[mi][4 + 2 * kk + 1][1]);
            ko.u[3] = pack2bf(acc[mi][4 + 2 * kk + 1][2], acc[mi][4 + 2 * kk + 1][3]);
            st = WIW_MFMA(ko.v, qo.v, st);
        }
        float sv4[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv4[r] = (fq * 4 + r) < p.T ? st[r] * p.scale_log2e : -INFINITY;
            mx = fmaxf(mx, sv4[r]);
        }
        mx = xor32_max(xor16_max(mx));
        float pr[4], l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pr[r] = __builtin_amdgcn_exp2f(sv4[r] - mx); l += pr[r]; }
        l = xor32_sum(xor16_sum(l));
        const float inv = __builtin_amdgcn_rcpf(l);
        union { uint32_t u[4]; bf16x8 v; } po;
        po.u[0] = pack2bf(pr[0], pr[1]); po.u[1] = pack2bf(pr[2], pr[3]); po.u[2] = 0u; po.u[3] = 0u;
        // v: lane = d (frow), rows = frames 4*fq + r -> needs the statistics of THOSE rows (held by lanes frow' = 4*fq + r)
        float rsv[4], nrv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rsv[r] = __shfl(rs, fq * 4 + r, 64);
            nrv[r] = __shfl(nrm, fq * 4 + r, 64);
        }
        // O^T = V^T . P^T per 16-d block; lane: query frow, d = 16*nb + 4*fq + r
        char* wrow = stg + (mi * TP + frow) * STG_ROWB + fq * 8;
#pragma unroll
        for (int nbk = 0; nbk < 4; ++nbk) {
            const float svv = fs[128 + nbk * 16 + frow], tvv = fs[W_ROWS + 128 + nbk * 16 + frow];
            union { uint32_t u[4]; bf16x8 v; } vo;
#if WIW_T_EPI == 1
            const wiw_f32x2 sv2 = {svv, svv}, tv2 = {tvv, tvv};
            const wiw_f32x2 v01 = pk_fma(wiw_f32x2{acc[mi][8 + nbk][0], acc[mi][8 + nbk][1]}, wiw_f32x2{rsv[0], rsv[1]},
                                         pk_fma(wiw_f32x2{nrv[0], nrv[1]}, sv2, tv2));
            const wiw_f32x2 v23 = pk_fma(wiw_f32x2{acc[mi][8 + nbk][2], acc[mi][8 + nbk][3]}, wiw_f32x2{rsv[2], rsv[3]},
                                         pk_fma(wiw_f32x2{nrv[2], nrv[3]}, sv2, tv2));
            vo.u[0] = pack2bf(v01.x, v01.y); vo.u[1] = pack2bf(v23.x, v23.y); vo.u[2] = 0u; vo.u[3] = 0u;
#else
            float vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vv[r] = __builtin_fmaf(acc[mi][8 + nbk][r], rsv[r], __builtin_fmaf(nrv[r], svv, tvv));
            vo.u[0] = pack2bf(vv[0], vv[1]); vo.u[1] = pack2bf(vv[2], vv[3]); vo.u[2] = 0u; vo.u[3] = 0u;
#endif
            f32x4 ot = f32x4{0.f, 0.f, 0.f, 0.f};
            ot = WIW_MFMA(vo.v, po.v, ot);
            uint2 pk;
#if WIW_T_EPI == 1
            const wiw_f32x2 inv2 = {inv, inv};
            const wiw_f32x2 o01 = wiw_f32x2{ot[0], ot[1]} * inv2, o23 = wiw_f32x2{ot[2], ot[3]} * inv2;
            pk.x = pack2bf(o01.x, o01.y);
            pk.y = pack2bf(o23.x, o23.y);
#else
            pk.x = pack2bf(ot[0] * inv, ot[1] * inv);
            pk.y = pack2bf(ot[2] * inv, ot[3] * inv);
#endif
            if (frow < TP) *(uint2*)(wrow + nbk * 32) = pk;
        }
    }
    wave_lds_sync();
    // row-major: 8 lanes x 16 B = one 128-byte row of O; NST unconditional store instructions.  All rows are read into
    // their OWN registers first: a store whose data registers are re-used by the next LDS read makes the compiler wait
    // `vmcnt(0)` for it — and with it for the next item's A prefetch, a full memory latency per store (r06j trace).
    // 32-bit byte offsets (the launcher checks O < 4 GiB).
    {
        uint4* dump = t_dump + (blockIdx.x & 511) * 64 + lane;
        uint4 vrow[NST];
#pragma unroll
        for (int k = 0; k < NST; ++k) vrow[k] = *(const uint4*)(stg + (k * 8 + rsub) * STG_ROWB + (lane & 7) * 16);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int sg = k * 8 + rsub;                  // staged row: (site mi, frame)
            const int sl = sg / TP, fr = sg - sl * TP;    // sl = 0, 1 (>= 2: beyond the wave's rows)
            const int site = s0_cur + wave * 2 + sl;
            const bool ok = sl < 2 && fr < p.T && site < p.S;
            const uint32_t m = (uint32_t)((b_cur * p.T + fr) * p.S + site);
            const uint32_t off = (m * (uint32_t)p.ldo + (uint32_t)(h_cur * 64 + (lane & 7) * 8)) * 2u;
            uint4* dst = ok ? (uint4*)((char*)p.O + off) : dump;
            *dst = vrow[k];
        }
    }
    wave_lds_sync();
}

// Row moments out of the Gram accumulators (WIW_T_MFMA_STATS): per-lane PARTIALS in the form attention_epilogue reduces (it
// sums over the four fq lanes of a row) — the lane that holds the moment contributes it, the others zero.
WIW_DEV void stats_from_gram(const f32x4 (&s1)[2], const f32x4 (&g)[2], float (&sum1)[2], float (&sum2)[2], int lane) {
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = frow & 3;
        const float d = r == 0 ? g[mi][0] : (r == 1 ? g[mi][1] : (r == 2 ? g[mi][2] : g[mi][3]));
        sum2[mi] = fq == (frow >> 2) ? d : 0.f;
        sum1[mi] = fq == 0 ? s1[mi][0] : 0.f;
    }
}
WIW_DEV void sq_from_gram(const f32x4 (&g)[2], float (&sum2)[2], int lane) {
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = frow & 3;
        const float d = r == 0 ? g[mi][0] : (r == 1 ? g[mi][1] : (r == 2 ? g[mi][2] : g[mi][3]));
        sum2[mi] = fq == (frow >> 2) ? d : 0.f;
    }
}
WIW_DEV bf16x8 ones_fragment() {
    union { uint32_t u[4]; bf16x8 v; } one;
    one.u[0] = one.u[1] = one.u[2] = one.u[3] = WIW_ONE16 * 0x10001u;
    return one.v;
}

// ---- RING form (any C; used for C > 320): A and W tiles through a 3-stage LDS ring, two wave groups one slot apart.
// (Round 3 tried A global -> VGPR and the VMEM instructions between the MFMAs for this form as well: 248 / 220 us at
// C = 640 / 1280 against 219 / 194 us for this one — a wave's VMEM issue (~110 cycles per instruction) serialises with ITS
// OWN MFMA issue, and here the other wave group's MFMA slot hides it; profiles/r06_temporal_traces.txt)
template <int TP>   // frame slots staged per site (T rounded up: 4, 8 or 14); rows of frames >= T read zeros
__global__ __launch_bounds__(NW * 64, 2) void temporal_block_kernel(const TemporalArgs p) {
    constexpr int A_ROWS = 16 * TP;
    constexpr int A_BYTES = A_ROWS * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int A_INSTR = 2 * TP;                    // 8-row DMA instructions per A tile
    constexpr int A_FULL = A_INSTR / NW;               // per wave
    constexpr bool A_HALF = (A_INSTR % NW) != 0;       // plus one 4-row instruction per wave (lanes 0..31)
    static_assert(A_INSTR % NW == 0 || A_INSTR % NW == 4, "A tile must split into whole + half instructions");
    constexpr int W_FULL = 3;                          // 24 instructions / 8 waves
    constexpr int LPT = A_FULL + (A_HALF ? 1 : 0) + W_FULL;
    constexpr int P3 = 1;                              // DMA instructions of part 3 (issued after the tile-(kt+1) wait)
    constexpr int NST = (2 * TP + 7) / 8;              // O store instructions per wave per item
    constexpr int STG_WAVE = NST * 8 * STG_ROWB;
    static_assert(NW * STG_WAVE <= STAGE_BYTES, "per-wave O staging must fit in one ring stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(lptr_t)smem;
    char* const scratch = smem + STAGES * STAGE_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int rsub = lane >> 3;   // DMA row of this lane inside an 8-row instruction

    // ---- item schedule: item = m_tile * heads + h (h fastest).  Blocks are renumbered so that the blocks of one XCD
    // (observed: block b runs on XCD b % 8) hold CONSECUTIVE items: the `heads` items of an m-tile run on one XCD at
    // about the same time and its A rows enter that L2 once.  Placement changes speed only.
    const int nb = gridDim.x;
    int lb = blockIdx.x;
    if ((nb & 7) == 0) lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
    int item = lb;
    if (item >= p.items) return;

    // per-lane source BYTE OFFSETS (32-bit: the tensors are < 4 GiB) from the wave-uniform bases X / W / fold
    uint32_t a_ofs[A_FULL + 1];
    bool a_ok[A_FULL + 1];
    uint32_t w_ofs[W_FULL];
    uint32_t f_ofs;
    const char* const Xb = (const char*)p.X;
    const char* const Wb = (const char*)p.W;
    const char* const Fb = (const char*)p.fold;
    int ld_kt = 0;
    int b_cur = 0, s0_cur = 0, h_cur = 0;

    auto setup_loader = [&](int it) {
        const int h = it % p.heads, mt = it / p.heads;
        const int b = mt / p.tiles_per_batch, s0 = (mt - b * p.tiles_per_batch) * 16;
#pragma unroll
        for (int i = 0; i < A_FULL + 1; ++i) {
            int rho;       // LDS row of this lane in instruction i
            if (i < A_FULL) rho = (wave * A_FULL + i) * 8 + rsub;
            else rho = A_FULL * NW * 8 + wave * 4 + (rsub & 3);
            const int site = rho / TP, fr = rho - site * TP;
            const int chunk = (lane & 7) ^ (rho & 7);
            const bool ok = fr < p.T && s0 + site < p.S && rho < A_ROWS;
            a_ok[i] = ok;
            const int64_t m = ((int64_t)b * p.T + fr) * p.S + s0 + site;
            a_ofs[i] = ok ? (uint32_t)((m * p.C + chunk * 8) * 2) : 0u;
        }
#pragma unroll
        for (int i = 0; i < W_FULL; ++i) {
            // Wqkv is TILED by the host: 1-KiB blocks [row / 8][k tile][8 rows x 128 B, chunks pre-swizzled] — one DMA
            // instruction reads one contiguous KiB (63 instead of 25 B/clk/CU, tools/ubench/lds_fill.hip)
            const int blk = h * (W_ROWS / 8) + wave * W_FULL + i;
            w_ofs[i] = (uint32_t)((int64_t)blk * (p.C / BK) * 1024 + lane * 16);
        }
        f_ofs = (uint32_t)(h * 2048 + (wave & 1) * 1024 + lane * 16);
    };
    auto a_src = [&](int i) -> const char* { return a_ok[i] ? Xb + (a_ofs[i] + (uint32_t)ld_kt * (BK * 2)) : p.zeros; };
    auto w_src = [&](int i) -> const char* { return Wb + (w_ofs[i] + (uint32_t)ld_kt * 1024u); };

    // all DMA instructions of the loader's next K tile (prologue / cross-item prefetch)
    auto issue_next = [&](int stage) {
        char* sA = smem + stage * STAGE_BYTES + wave * A_FULL * 1024;
        char* sW = smem + stage * STAGE_BYTES + A_BYTES + wave * W_FULL * 1024;
#pragma unroll
        for (int i = 0; i < A_FULL; ++i) glds16p(a_src(i), LDSA(sA + i * 1024));
        if (A_HALF) {
            char* sH = smem + stage * STAGE_BYTES + A_FULL * NW * 1024 + wave * 512;
            if (lane < 32) glds16p(a_src(A_FULL), LDSA(sH));
        }
#pragma unroll
        for (int i = 0; i < W_FULL; ++i) glds16p(w_src(i), LDSA(sW + i * 1024));
        ++ld_kt;
    };
    // the same instructions spread over the four slots of the previous K tile: (A..) | (A.. + half) | (W0 W1) | (W2)
    auto issue_part = [&](int stage, auto part_tag) {
        constexpr int part = decltype(part_tag)::value;
        char* sA = smem + stage * STAGE_BYTES + wave * A_FULL * 1024;
        char* sW = smem + stage * STAGE_BYTES + A_BYTES + wave * W_FULL * 1024;
        constexpr int A0 = (A_FULL + 1) / 2;   // full A instructions in part 0
        if (part == 0) {
#pragma unroll
            for (int i = 0; i < A0; ++i) glds16p(a_src(i), LDSA(sA + i * 1024));
        } else if (part == 1) {
#pragma unroll
            for (int i = A0; i < A_FULL; ++i) glds16p(a_src(i), LDSA(sA + i * 1024));
            if (A_HALF) {
                char* sH = smem + stage * STAGE_BYTES + A_FULL * NW * 1024 + wave * 512;
                if (lane < 32) glds16p(a_src(A_FULL), LDSA(sH));
            }
        } else if (part == 2) {
            glds16p(w_src(0), LDSA(sW));
            glds16p(w_src(1), LDSA(sW + 1024));
        } else {
            glds16p(w_src(2), LDSA(sW + 2048));
            ++ld_kt;
        }
    };

    // ---- fragment addressing.  A rows of this wave: site mi -> LDS rows wave*2*TP + mi*TP + min(frow, TP-1)
    int a_off[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int rho = wave * 2 * TP + mi * TP + (frow < TP ? frow : TP - 1);
        a_off[mi] = rho * 128 + ((fq ^ (rho & 7)) << 4);        // k-step 0; k-step 1 flips bit 2 of the chunk: ^ 64
    }
#if WIW_T_MFMA_STATS
    f32x4 acc_s1[2], acc_s2[2];
    const bf16x8 ones = ones_fragment();
#endif
    const int w_off = frow * 128 + ((fq ^ (frow & 7)) << 4);

    f32x4 acc[2][NF];
    bf16x8 fa[2], fb[NF];
    float sum1[2], sum2[2];
    auto read_frags = [&](int stage, int kk) {
        const char* sA = smem + stage * STAGE_BYTES;
        const char* sW = sA + A_BYTES;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const bf16x8*)(sA + (a_off[mi] ^ (kk << 6)));
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) fb[ni] = *(const bf16x8*)(sW + ni * 2048 + (w_off ^ (kk << 6)));
    };
    auto mma = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)   // q, k: swapped operands -> lane = row frow, columns 16*ni + 4*fq + r
                acc[mi][ni] = WIW_MFMA(fb[ni], fa[mi], acc[mi][ni]);
#pragma unroll
        for (int ni = 8; ni < NF; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)   // v: lane = column (d) frow, rows (frames) 4*fq + r
                acc[mi][ni] = WIW_MFMA(fa[mi], fb[ni], acc[mi][ni]);
#if WIW_T_MFMA_STATS
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {          // the two row moments (see the top of the file)
            acc_s1[mi] = WIW_MFMA(ones, fa[mi], acc_s1[mi]);
            acc_s2[mi] = WIW_MFMA(fa[mi], fa[mi], acc_s2[mi]);
        }
        __builtin_amdgcn_s_setprio(0);
        return;
#endif
        // LayerNorm statistics from the A fragments just consumed: one v_dot2c per packed pair and moment (16 VALU
        // instructions per k-step), each issued in the 16-cycle shadow of an MFMA.  (Round 2 unpacked to fp32 and ran
        // 48 instructions AFTER the last MFMA: the wave issues in order, so they extended every MFMA slot by ~260 cycles.)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            union { bf16x8 v; uint32_t u[4]; } x;
            x.v = fa[mi];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sum1[mi] = dot2_acc(x.u[j], WIW_ONE16 * 0x10001u, sum1[mi]);
                sum2[mi] = dot2_acc(x.u[j], x.u[j], sum2[mi]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto slot_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    const bool lag = wave >= NW / 2;
    const int nk = p.C / BK;

    // ---- prologue
    setup_loader(item);
    ld_kt = 0;
    int st_c = 0;
    int parity = 0;                  // fold-vector scratch buffer of the current item
    int pending_stores = 0;
    glds16p(Fb + f_ofs, LDSA(scratch + parity * SCR_BYTES + (wave & 1) * 1024));
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < nk) issue_next(j);

    while (item >= 0) {
        {
            const int h = item % p.heads, mt = item / p.heads;
            h_cur = h;
            b_cur = mt / p.tiles_per_batch;
            s0_cur = (mt - b_cur * p.tiles_per_batch) * 16;
        }
        setup_loader(item);   // loader pointers of THIS item (recomputed: not kept live across the epilogue)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            sum1[mi] = 0.f; sum2[mi] = 0.f;
#if WIW_T_MFMA_STATS
            acc_s1[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_s2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // wait until this wave's DMA portion of K tile j has landed; n_dma = DMA instructions issued after it
        // (vmcnt retires in order; the previous item's O stores queue behind the prefetched tiles 0 and 1)
        auto wait_tile = [&](int j, auto ndma_tag) {
            constexpr int n_dma = decltype(ndma_tag)::value;
            const bool dma_younger = j + 1 < nk;
            const bool st = (j < D) && pending_stores != 0;
            if (dma_younger) {
                if (st) wait_vmcnt<n_dma + NST>();
                else wait_vmcnt<n_dma>();
            } else {
                if (st) wait_vmcnt<NST>();
                else wait_vmcnt<0>();
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        using ILPT = std::integral_constant<int, LPT>; using ILATE [[maybe_unused]] = std::integral_constant<int, LPT - P3>;
        wait_tile(0, ILPT{});
        if (lag) slot_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            slot_barrier();                                  // local barrier 4kt
            const bool more = kt + D < nk;
            int si = st_c + D;
            si = si >= STAGES ? si - STAGES : si;
            // all DMA instructions of K tile kt + D in ONE slot: a slot that issues ANY LDS-DMA pays ~180 cycles once, further
            // instructions ~25 each (tools/trace_probe.py)
            if (more) { issue_part(si, I0{}); issue_part(si, I1{}); issue_part(si, I2{}); issue_part(si, I3{}); }
            read_frags(st_c, 0);
            slot_barrier();                                  // 4kt+1
            mma();
            slot_barrier();                                  // 4kt+2
            read_frags(st_c, 1);
            if (kt + 1 < nk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wait_tile(kt + 1, ILPT{});
            }
            slot_barrier();                                  // 4kt+3
            mma();
            st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
        }
        if (!lag) slot_barrier();   // leading group: the lagging group has finished reading the ring

        // ---- keep the DMA stream going: fold vectors + first D K tiles of the NEXT item -> stages st_c, st_c + 1
        const int item_next = item + nb < p.items ? item + nb : -1;
        if (item_next >= 0) {
            setup_loader(item_next);
            ld_kt = 0;
            glds16p(Fb + f_ofs, LDSA(scratch + (parity ^ 1) * SCR_BYTES + (wave & 1) * 1024));
#pragma unroll
            for (int j = 0; j < D; ++j) {
                if (j < nk) {
                    int sj = st_c + j;
                    sj = sj >= STAGES ? sj - STAGES : sj;
                    issue_next(sj);
                }
            }
        }
        int st_e = st_c + D;
        st_e = st_e >= STAGES ? st_e - STAGES : st_e;

#if WIW_T_MFMA_STATS
        stats_from_gram(acc_s1, acc_s2, sum1, sum2, lane);
#endif
        // ---- epilogue (per wave, no block barrier); O staging in the ring stage nobody reads or fills now
        attention_epilogue<TP>(acc, sum1, sum2, (const float*)(scratch + parity * SCR_BYTES),
                               smem + st_e * STAGE_BYTES + wave * STG_WAVE, p, b_cur, s0_cur, h_cur, wave, lane);
        pending_stores = NST;
        parity ^= 1;
        item = item_next;
    }
}

// ---- weight-RESIDENT form for C <= 320 (the UNet's first level: 5 heads, 258 048 rows — the largest temporal block).
// The q|k|v weights of ONE head are 192 x C x 2 B <= 120 KiB: they fit in LDS next to the fold vectors (2 KiB) and the O
// staging (36 KiB).  A block keeps its head for the whole launch and walks over site tiles; with A going global -> VGPR
// (each wave's rows are private) NOTHING is written to LDS after the prologue, so there is no ring, no slot barrier and no
// wave group: eight waves free-run, and while one wave of a SIMD is in its VALU-bound attention epilogue the other one
// keeps the matrix pipe busy (the ring form above spends ~40 % of an item in epilogue + barrier skew with the pipe idle,
// and its slots are as long as the wave's VMEM ISSUE time: ~110 cycles per instruction, profiles/r06*_temporal_trace.txt).
//   * grid = 8 XCDs x floor(32 / heads) tile slots x heads blocks, renumbered so that the `heads` blocks of a tile slot sit on
//     one XCD and read its A rows from that L2 together; slot j handles tiles j, j + nslots, ...
//   * per k-step a wave reads its 12 W fragments in two halves, each half re-filled (for the NEXT k-step) right behind the
//     12 MFMAs that consumed it: the reads of one half fly under the MFMAs of the other.
template <int TP>
__global__ __launch_bounds__(NW * 64, 2) void temporal_block_resident_kernel(const TemporalArgs p, const int nslots) {
    constexpr int NST = (2 * TP + 7) / 8;
    constexpr int STG_WAVE = NST * 8 * STG_ROWB;
    constexpr int A_LD = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int nk = p.C / BK;
    char* const scratch = smem + nk * W_BYTES;
    char* const ostage = scratch + SCR_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int nb = gridDim.x;
    int lb = blockIdx.x;
    if ((nb & 7) == 0) lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
    const int h = lb % p.heads;
    int mt = lb / p.heads;
    const int m_tiles = p.batch * p.tiles_per_batch;
    if (mt >= m_tiles) return;
    const char* const Xb = (const char*)p.X;

    // ---- prologue: W_h (host-tiled 1-KiB blocks [row / 8][k tile]) -> LDS [k tile][row / 8], fold vectors
    {
        const uint32_t hbase = (uint32_t)h * (W_ROWS / 8) * (uint32_t)nk * 1024u;
        for (int i = wave; i < (W_ROWS / 8) * nk; i += NW) {
            const int kt = i / (W_ROWS / 8), rb = i - kt * (W_ROWS / 8);
            glds16((const char*)p.W, hbase + (uint32_t)(rb * nk + kt) * 1024u + lane * 16, LDSA(smem + i * 1024));
        }
        if (wave < 2) glds16((const char*)p.fold, (uint32_t)(h * 2048 + wave * 1024 + lane * 16), LDSA(scratch + wave * 1024));
        wait_vmcnt<0>();
        __syncthreads();
    }
    const float* const fs = (const float*)scratch;
#ifdef WIW_T_TRACE
    char* const rt_base = ostage + NW * STG_WAVE;   // 1 KiB behind the kernel's own LDS
    int r_ord = 0;
#define RTP(idx) do { if (blockIdx.x == 0 && lane == 0 && r_ord == TR_ITEM) *(volatile __attribute__((address_space(3))) long long*)(lptr_t)(rt_base + (wave * 16 + (idx)) * 8) = __builtin_readcyclecounter(); } while (0)
#else
#define RTP(idx) do { } while (0)
#endif

    uint32_t a_ofs[2], a_nxt[2];
    auto row_ofs = [&](int mtile, uint32_t (&o)[2]) {   // see setup_loader of the ring form
        const int b = mtile / p.tiles_per_batch, s0 = (mtile - b * p.tiles_per_batch) * 16;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            int site = s0 + wave * 2 + mi;
            site = site < p.S ? site : p.S - 1;
            const int fr = frow < p.T ? frow : p.T - 1;
            const int64_t m = ((int64_t)b * p.T + fr) * p.S + site;
            o[mi] = (uint32_t)((m * p.C + fq * 8) * 2);
        }
    };
    auto load_a = [&](bf16x8 (&dst)[2][2], const uint32_t (&o)[2], int kt) {   // asm: see load_a_one of the ring form
        const char* const xb = Xb;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const uint32_t vo = o[mi] + (uint32_t)(kt * (BK * 2));
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0][mi]) : "v"(vo), "s"(xb) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(dst[1][mi]) : "v"(vo), "s"(xb) : "memory");
        }
    };
#if WIW_T_MFMA_STATS
    f32x4 acc_s2[2];
#if WIW_T_RES_S1_MFMA
    f32x4 acc_s1[2];
    const bf16x8 ones = ones_fragment();
#endif
#endif

    const int w_off = frow * 128 + ((fq ^ (frow & 7)) << 4);
    f32x4 acc[2][NF];
    bf16x8 fa[2][2][2];   // [buffer][k-step][site]
    bf16x8 fb[NF];
    float sum1[2], sum2[2];
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    // W fragments of k-step `step` (= 2 * k tile + k-step), half hf: fragments 6*hf .. 6*hf + 5
    auto read_half = [&](int step, auto half_tag) {
        constexpr int hf = decltype(half_tag)::value;
        const char* sW = smem + (step >> 1) * W_BYTES;
        const int sw = w_off ^ ((step & 1) << 6);
#pragma unroll
        for (int ni = hf * 6; ni < hf * 6 + 6; ++ni) {
#if WIW_T_ABLATE & 4
            if (step == 0) fb[ni] = *(const bf16x8*)(sW + ni * 2048 + sw);
            asm volatile("" : "+v"(fb[ni]));
#else
            fb[ni] = *(const bf16x8*)(sW + ni * 2048 + sw);
#endif
        }
    };
    // 12 MFMAs of one half + the LayerNorm statistics of site hf (8 v_dot2c) in their shadow
    auto mma_half = [&](const bf16x8 (&a)[2], auto half_tag) {
        constexpr int hf = decltype(half_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = hf * 6; ni < hf * 6 + 6; ++ni) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (ni < 8) acc[mi][ni] = WIW_MFMA(fb[ni], a[mi], acc[mi][ni]);   // q, k: swapped operands
                else acc[mi][ni] = WIW_MFMA(a[mi], fb[ni], acc[mi][ni]);          // v
            }
        }
#if WIW_T_MFMA_STATS && !(WIW_T_ABLATE & 8)
        acc_s2[hf] = WIW_MFMA(a[hf], a[hf], acc_s2[hf]);   // row sums of squares of site hf: the diagonal of A . A^T
#if WIW_T_RES_S1_MFMA
        acc_s1[hf] = WIW_MFMA(ones, a[hf], acc_s1[hf]);
#else
        {
            union { bf16x8 v; uint32_t u[4]; } x;
            x.v = a[hf];
#pragma unroll
            for (int j = 0; j < 4; ++j) sum1[hf] = dot2_acc(x.u[j], WIW_ONE16 * 0x10001u, sum1[hf]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        }
#endif
#elif !(WIW_T_ABLATE & 8)
        union { bf16x8 v; uint32_t u[4]; } x;
        x.v = a[hf];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sum1[hf] = dot2_acc(x.u[j], WIW_ONE16 * 0x10001u, sum1[hf]);
            sum2[hf] = dot2_acc(x.u[j], x.u[j], sum2[hf]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // 1 VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#endif
        asm volatile("" : "+v"(sum1[hf]), "+v"(sum2[hf]));
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    row_ofs(mt, a_ofs);
    load_a(fa[0], a_ofs, 0);
    read_half(0, I0{});
    read_half(0, I1{});
    int pending_stores = 0;
    const int nsteps = 2 * nk;
    // one item; START = the A buffer its K tile 0 lives in (compile time: a run-time buffer index costs ~300 spilled
    // registers).  An odd number of K tiles flips the start buffer from item to item.  Returns false after the last item.
    auto item_body = [&](auto start_tag) -> bool {
        constexpr int START = decltype(start_tag)::value;
        const int mt_next = mt + nslots;
        const bool has_next = mt_next < m_tiles;
        if (has_next) row_ofs(mt_next, a_nxt);
        RTP(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            sum1[mi] = 0.f; sum2[mi] = 0.f;
#if WIW_T_MFMA_STATS
#if WIW_T_RES_S1_MFMA
            acc_s1[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
            acc_s2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto ktile = [&](auto buf_tag, int kt) {
            constexpr int P = decltype(buf_tag)::value;
            RTP(1 + 2 * kt);
            // the next A tile (of this item, or tile 0 of the next one) into the other buffer, then wait for THIS tile:
            // younger in the vmcnt queue are those loads and, at kt = 0, the previous item's O stores
            int younger = kt == 0 ? pending_stores : 0;
#if WIW_T_ABLATE & 2   // timing only: no A loads after the first tile (stale operands)
            if (false) load_a(fa[P ^ 1], a_ofs, kt + 1);
#else
            if (kt + 1 < nk) { load_a(fa[P ^ 1], a_ofs, kt + 1); younger += A_LD; }
            else if (has_next) { load_a(fa[P ^ 1], a_nxt, 0); younger += A_LD; }
#endif
            wait_vmcnt_rt(younger);
            asm volatile("" : "+v"(fa[P][0][0]), "+v"(fa[P][0][1]), "+v"(fa[P][1][0]), "+v"(fa[P][1][1]));
            RTP(2 + 2 * kt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int step = 2 * kt + kk;
                const int nxt = step + 1 == nsteps ? 0 : step + 1;
                mma_half(fa[P][kk], I0{});
                read_half(nxt, I0{});
                __builtin_amdgcn_sched_barrier(0);
                mma_half(fa[P][kk], I1{});
                read_half(nxt, I1{});
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(std::integral_constant<int, START>{}, kt);
            if (kt + 1 < nk) ktile(std::integral_constant<int, START ^ 1>{}, kt + 1);
        }
        RTP(11);
#if WIW_T_ABLATE & 1   // timing only: no attention epilogue (the accumulators are kept alive by an empty asm)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
        asm volatile("" ::"v"(sum1[0]), "v"(sum1[1]), "v"(sum2[0]), "v"(sum2[1]));
#else
        const int b_cur = mt / p.tiles_per_batch, s0_cur = (mt - b_cur * p.tiles_per_batch) * 16;
#if WIW_T_MFMA_STATS
#if WIW_T_RES_S1_MFMA
        stats_from_gram(acc_s1, acc_s2, sum1, sum2, lane);
#else
        sq_from_gram(acc_s2, sum2, lane);           // sum1 already holds the lanes' v_dot2c partials
#endif
#endif
        attention_epilogue<TP>(acc, sum1, sum2, fs, ostage + wave * STG_WAVE, p, b_cur, s0_cur, h, wave, lane);
#endif
        RTP(12);
#ifdef WIW_T_TRACE
        if (blockIdx.x == 0 && lane == 0 && r_ord == TR_ITEM)
            for (int i = 0; i < 16; ++i)
                g_rtrace[wave][i] = *(volatile __attribute__((address_space(3))) long long*)(lptr_t)(rt_base + (wave * 16 + i) * 8);
        ++r_ord;
#endif
#if WIW_T_ABLATE & 1
        pending_stores = 0;
#else
        pending_stores = NST;
#endif
        mt = mt_next;
        a_ofs[0] = a_nxt[0]; a_ofs[1] = a_nxt[1];
        return has_next;
    };
    // Waves w and w + 4 share a SIMD and run IDENTICAL work: started together they stay in lock-step — both in the MFMA
    // phase (each gets half the matrix pipe), then both in the VALU-bound epilogue (pipe idle).  Half an item of head start
    // for one of them makes the phases complementary, and complementary phases do not contend, so the offset persists.
    // (~512 cycles per s_sleep 8; half an item ~ (nk + 5) of them: ~5 000 cycles at C = 320, once per launch)
#if WIW_T_STAGGER
    if (wave >= NW / 2)
        for (int i = 0; i < nk + 5; ++i) __builtin_amdgcn_s_sleep(8);
#endif
    const bool flip = (nk & 1) != 0;
    while (true) {
        if (!item_body(I0{})) break;
        if (flip && !item_body(I1{})) break;
    }
}

template <int TP>
int launch_tp(hipStream_t s, const TemporalArgs& a) {
    constexpr int SMEM = STAGES * (16 * TP * BK * 2 + W_BYTES) + 2 * SCR_BYTES;
    static std::once_flag once;
    static bool attr_ok = false;
    static int num_cu = 256;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute((const void*)temporal_block_kernel<TP>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) == hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
    });
    if (!attr_ok) {
        wiw_set_error("hipFuncSetAttribute(temporal_block) failed");
        return WIW_ELAUNCH;
    }
    // C <= 320: one head's weights stay in LDS (see temporal_block_resident_kernel); WIW_TEMPORAL_RING=1 forces the ring form
    constexpr int SMEM_R_FIXED = SCR_BYTES + NW * ((2 * TP + 7) / 8) * 8 * STG_ROWB;
#ifdef WIW_T_TRACE
    const int smem_r = (a.C / BK) * W_BYTES + SMEM_R_FIXED + 1024;
#else
    const int smem_r = (a.C / BK) * W_BYTES + SMEM_R_FIXED;
#endif
    const int per_xcd = (num_cu / 8) / a.heads;
    static const bool force_ring = getenv("WIW_TEMPORAL_RING") != nullptr;
    if (!force_ring && smem_r <= 160 * 1024 && per_xcd >= 1 && (num_cu & 7) == 0) {
        static std::once_flag once_r;
        static bool attr_r = false;
        std::call_once(once_r, [] {
            attr_r = hipFuncSetAttribute((const void*)temporal_block_resident_kernel<TP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024) == hipSuccess;
        });
        if (!attr_r) {
            wiw_set_error("hipFuncSetAttribute(temporal_block_resident) failed");
            return WIW_ELAUNCH;
        }
        const int nslots = 8 * per_xcd;
        hipLaunchKernelGGL((temporal_block_resident_kernel<TP>), dim3((unsigned)(nslots * a.heads)), dim3(NW * 64), smem_r, s, a, nslots);
        return wiw_check_launch("wiw_temporal_attn_block_bf16");
    }
    int grid = a.items < num_cu ? a.items : num_cu;
    if (grid >= 64) grid &= ~7;   // keep the XCD renumbering bijective
    hipLaunchKernelGGL((temporal_block_kernel<TP>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, a);
    return wiw_check_launch("wiw_temporal_attn_block_bf16");
}

}  // namespace

#ifdef WIW_T_TRACE
extern "C" int wiw_temporal_rtrace_read(long long* out) {   // debug builds only: resident form, [8 waves][16]
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rtrace), sizeof(long long) * 8 * 16) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int wiw_temporal_attn_block_bf16(void* stream, const void* X, const void* Wqkv, const float* fold, void* O,
                                            int ldo, int batch, int T, int S, int heads, float eps, float scale,
                                            const void* zeros) {
    WIW_REQUIRE(X && Wqkv && fold && O && zeros, "temporal_attn_block: null pointer");
    WIW_REQUIRE(batch > 0 && S > 0 && heads > 0, "temporal_attn_block: bad sizes");
    WIW_REQUIRE(T >= 1 && T <= 14, "temporal_attn_block: 1 <= T <= 14 (use wiw_attn_temporal_bf16 beyond)");
    WIW_REQUIRE(ldo % 8 == 0 && ldo >= heads * 64, "temporal_attn_block: ldo must be a multiple of 8 and >= C");
    WIW_REQUIRE((((uintptr_t)X | (uintptr_t)Wqkv | (uintptr_t)fold | (uintptr_t)O | (uintptr_t)zeros) & 15) == 0,
                "temporal_attn_block: pointers must be 16-byte aligned");
    TemporalArgs a;
    a.X = (const uint16_t*)X; a.W = (const uint16_t*)Wqkv; a.fold = fold; a.O = (uint16_t*)O; a.zeros = (const char*)zeros;
    a.ldo = ldo; a.batch = batch; a.T = T; a.S = S; a.C = heads * 64; a.heads = heads;
    a.tiles_per_batch = (S + 15) / 16;
    const int64_t items = (int64_t)batch * a.tiles_per_batch * heads;
    WIW_REQUIRE(items < (1ll << 31), "temporal_attn_block: too many work items");
    WIW_REQUIRE((int64_t)batch * T * S * heads * 64 * 2 < (1ll << 32), "temporal_attn_block: X must be < 4 GiB (32-bit source offsets)");
    WIW_REQUIRE((int64_t)batch * T * S * ldo * 2 < (1ll << 32), "temporal_attn_block: O must be < 4 GiB (32-bit destination offsets)");
    a.items = (int)items;
    a.eps = eps; a.scale_log2e = scale * LOG2E; a.inv_c = 1.0f / (float)a.C;
    hipStream_t s = (hipStream_t)stream;
    if (T <= 4) return launch_tp<4>(s, a);
    if (T <= 8) return launch_tp<8>(s, a);
    return launch_tp<14>(s, a);
}
