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

#ifndef WIW_DMA_BURST
#define WIW_DMA_BURST 1   // 1: all DMA instructions of a K tile are issued in ONE slot (0: spread over the four slots)
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
WIW_DEV void glds16(const char* g, char* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }
template <int N>
WIW_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
WIW_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sink of the O stores of lanes without a valid row (the store count per wave stays a compile-time constant)
__device__ uint4 t_dump[512 * 64];

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
    char* const scratch = smem + STAGES * STAGE_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int rsub = lane >> 3;

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
        for (int i = 0; i < A_FULL; ++i) glds16(a_src(i), sA + i * 1024);
        if (A_HALF) {
            char* sH = smem + stage * STAGE_BYTES + A_FULL * NW * 1024 + wave * 512;
            if (lane < 32) glds16(a_src(A_FULL), sH);
        }
#pragma unroll
        for (int i = 0; i < W_FULL; ++i) glds16(w_src(i), sW + i * 1024);
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
            for (int i = 0; i < A0; ++i) glds16(a_src(i), sA + i * 1024);
        } else if (part == 1) {
#pragma unroll
            for (int i = A0; i < A_FULL; ++i) glds16(a_src(i), sA + i * 1024);
            if (A_HALF) {
                char* sH = smem + stage * STAGE_BYTES + A_FULL * NW * 1024 + wave * 512;
                if (lane < 32) glds16(a_src(A_FULL), sH);
            }
        } else if (part == 2) {
            glds16(w_src(0), sW);
            glds16(w_src(1), sW + 1024);
        } else {
            glds16(w_src(2), sW + 2048);
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
        __builtin_amdgcn_s_setprio(0);
        // LayerNorm statistics from the A fragments just consumed (VALU, behind the MFMAs in flight)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            union { bf16x8 v; uint32_t u[4]; } x;
            x.v = fa[mi];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const wiw_f32x2 lh = unpack2(x.u[j]);
                const float lo = lh.x, hi = lh.y;
                sum1[mi] += lo + hi;
                sum2[mi] = __builtin_fmaf(lo, lo, __builtin_fmaf(hi, hi, sum2[mi]));
            }
        }
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
    glds16(Fb + f_ofs, scratch + parity * SCR_BYTES + (wave & 1) * 1024);
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
#if WIW_DMA_BURST
            // a slot that issues ANY LDS-DMA pays ~180 cycles once, further instructions ~25 each (tools/trace_probe.py)
            if (more) { issue_part(si, I0{}); issue_part(si, I1{}); issue_part(si, I2{}); issue_part(si, I3{}); }
#else
            if (more) issue_part(si, I0{});
#endif
            read_frags(st_c, 0);
            slot_barrier();                                  // 4kt+1
#if !WIW_DMA_BURST
            if (more) issue_part(si, I1{});
#endif
            mma();
            slot_barrier();                                  // 4kt+2
#if !WIW_DMA_BURST
            if (more) issue_part(si, I2{});
#endif
            read_frags(st_c, 1);
            if (kt + 1 < nk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if WIW_DMA_BURST
                wait_tile(kt + 1, ILPT{});
#else
                wait_tile(kt + 1, ILATE{});
#endif
            }
            slot_barrier();                                  // 4kt+3
#if !WIW_DMA_BURST
            if (more) issue_part(si, I3{});
#endif
            mma();
            st_c = (st_c + 1 == STAGES) ? 0 : st_c + 1;
        }
        if (!lag) slot_barrier();   // leading group: the lagging group has finished reading the ring

        // ---- keep the DMA stream going: fold vectors + first D K tiles of the NEXT item -> stages st_c, st_c + 1
        const int item_next = item + nb < p.items ? item + nb : -1;
        if (item_next >= 0) {
            setup_loader(item_next);
            ld_kt = 0;
            glds16(Fb + f_ofs, scratch + (parity ^ 1) * SCR_BYTES + (wave & 1) * 1024);
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

        // ---- epilogue (per wave, no block barrier)
        const float* fs = (const float*)(scratch + parity * SCR_BYTES);   // s[192] | t[192]
        char* stg = smem + st_e * STAGE_BYTES + wave * STG_WAVE;
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
                acc[mi][ni][0] = __builtin_fmaf(acc[mi][ni][0], rs, __builtin_fmaf(nrm, sv.x, tv.x));
                acc[mi][ni][1] = __builtin_fmaf(acc[mi][ni][1], rs, __builtin_fmaf(nrm, sv.y, tv.y));
                acc[mi][ni][2] = __builtin_fmaf(acc[mi][ni][2], rs, __builtin_fmaf(nrm, sv.z, tv.z));
                acc[mi][ni][3] = __builtin_fmaf(acc[mi][ni][3], rs, __builtin_fmaf(nrm, sv.w, tv.w));
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
                float vv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    vv[r] = __builtin_fmaf(acc[mi][8 + nbk][r], rsv[r], __builtin_fmaf(nrv[r], svv, tvv));
                union { uint32_t u[4]; bf16x8 v; } vo;
                vo.u[0] = pack2bf(vv[0], vv[1]); vo.u[1] = pack2bf(vv[2], vv[3]); vo.u[2] = 0u; vo.u[3] = 0u;
                f32x4 ot = f32x4{0.f, 0.f, 0.f, 0.f};
                ot = WIW_MFMA(vo.v, po.v, ot);
                uint2 pk;
                pk.x = pack2bf(ot[0] * inv, ot[1] * inv);
                pk.y = pack2bf(ot[2] * inv, ot[3] * inv);
                if (frow < TP) *(uint2*)(wrow + nbk * 32) = pk;
            }
        }
        wave_lds_sync();
        // row-major: 8 lanes x 16 B = one 128-byte row of O; NST unconditional store instructions
        {
            uint4* dump = t_dump + (blockIdx.x & 511) * 64 + lane;
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int sg = k * 8 + rsub;                  // staged row: (site mi, frame)
                const int sl = sg / TP, fr = sg - sl * TP;    // sl = 0, 1 (>= 2: beyond the wave's rows)
                const int site = s0_cur + wave * 2 + sl;
                const bool ok = sl < 2 && fr < p.T && site < p.S;
                const uint4 v = *(const uint4*)(stg + sg * STG_ROWB + (lane & 7) * 16);
                const int64_t m = ((int64_t)b_cur * p.T + fr) * p.S + site;
                uint4* dst = ok ? (uint4*)(p.O + m * p.ldo + h_cur * 64 + (lane & 7) * 8) : dump;
                *dst = v;
            }
        }
        wave_lds_sync();
        pending_stores = NST;
        parity ^= 1;
        item = item_next;
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
    int grid = a.items < num_cu ? a.items : num_cu;
    if (grid >= 64) grid &= ~7;   // keep the XCD renumbering bijective
    hipLaunchKernelGGL((temporal_block_kernel<TP>), dim3((unsigned)grid), dim3(NW * 64), SMEM, s, a);
    return wiw_check_launch("wiw_temporal_attn_block_bf16");
}

}  // namespace

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
    a.items = (int)items;
    a.eps = eps; a.scale_log2e = scale * LOG2E; a.inv_c = 1.0f / (float)a.C;
    hipStream_t s = (hipStream_t)stream;
    if (T <= 4) return launch_tp<4>(s, a);
    if (T <= 8) return launch_tp<8>(s, a);
    return launch_tp<14>(s, a);
}
