// Cross-attention over a FEW keys (2 <= P <= 8 conditioning tokens per candidate) for gfx950 (MI355X): the general form of
// the two cross-attentions of a transformer layer (BasicTransformerBlock.attn2, dp/models/attention.py:545-551, and
// TemporalBasicTransformerBlock.attn2, :740-743, with AttnProcessor2_0, attention_processor.py:2358-2391) for checkpoints
// trained with --num_past_obs > 1 (train_svd.py:359, 889-894; pipeline_stable_video_diffusion.py:500-508: one CLIP embedding
// per past observation).  With ONE key — every launcher of the reference — the softmax is 1 and the whole operator is a
// vector per candidate (unet.py); that closed form stays the served path.
//
//   O[m][h*64 + d] = sum_p softmax_p( Q[m][h*64 : h*64+64] . K[item(m)][p][h*64 : ...] * scale ) V[item(m)][p][h*64 + d]
//
// Rows m of one item (a CFG-batch entry: T * S consecutive rows in this build's token order, spatial AND temporal blocks)
// share P keys: K / V of an item are P x C x 2 tensors that live in REGISTERS (a thread = one 8-channel chunk of every key);
// Q rows stream through once, O rows are written once — an HBM-bound kernel.  A head is 8 consecutive lanes (64 channels),
// always inside one wave (row starts are multiples of C / 8 lanes, C / 8 a multiple of 8): the 64-term dot product is 8 FMAs
// per lane and three cross-lane steps per key; the softmax over <= 8 keys is in registers, fp32.
#include "common.h"

namespace {

constexpr int CA_MAXP = 8;

WIW_DEV float oct_sum(float v) {   // sum over the 8 aligned lanes of a head
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void cross_attn_fewkeys_kernel(const uint16_t* __restrict__ Q, int ldq, const uint16_t* __restrict__ K,
                                                                 const uint16_t* __restrict__ V, uint16_t* __restrict__ O, int ldo, int C,
                                                                 int P, int rows_per_item, int rows_per_block, float scale_log2e) {
    const int tid = threadIdx.x;
    const int chunks = C >> 3;                  // <= 256 (launcher)
    const int rp = 256 / chunks;
    const int ci = tid % chunks, rl = tid / chunks;
    const int item = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows_per_item) r1 = rows_per_item;
    if (rl >= rp) return;    // surplus threads (256 % chunks): whole 8-lane groups, never part of a live head
    uint4 kq[CA_MAXP], vq[CA_MAXP];
#pragma unroll
    for (int p = 0; p < CA_MAXP; ++p) {
        const int pp = p < P ? p : P - 1;
        kq[p] = *(const uint4*)(K + ((int64_t)item * P + pp) * C + ci * 8);
        vq[p] = *(const uint4*)(V + ((int64_t)item * P + pp) * C + ci * 8);
    }
    const int64_t base = (int64_t)item * rows_per_item;
    for (int r = r0 + rl; r < r1; r += rp) {   // the 8 lanes of a head share rl: they leave the loop together
        float q[8];
        unpack8(*(const uint4*)(Q + (base + r) * ldq + ci * 8), q);
        float s[CA_MAXP], mx = -INFINITY;
#pragma unroll
        for (int p = 0; p < CA_MAXP; ++p) {
            float k[8], d = 0.f;
            unpack8(kq[p], k);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = __builtin_fmaf(q[e], k[e], d);
            s[p] = p < P ? oct_sum(d) * scale_log2e : -INFINITY;
            mx = fmaxf(mx, s[p]);
        }
        float l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < CA_MAXP; ++p) {
            const float w = p < P ? __builtin_amdgcn_exp2f(s[p] - mx) : 0.f;
            float v[8];
            unpack8(vq[p], v);
            l += w;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(w, v[e], o[e]);
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= inv;
        *(uint4*)(O + (base + r) * ldo + ci * 8) = pack8(o);
    }
}

}  // namespace

extern "C" int wiw_cross_attn_fewkeys_bf16(void* stream, const void* Q, int ldq, const void* K, const void* V, void* O, int ldo,
                                           int64_t rows, int rows_per_item, int heads, int P, float scale) {
    WIW_REQUIRE(Q && K && V && O, "cross_attn_fewkeys: null pointer");
    WIW_REQUIRE(heads > 0 && heads * 64 <= 2048, "cross_attn_fewkeys: heads * 64 channels, at most 2048");
    WIW_REQUIRE(P >= 1 && P <= CA_MAXP, "cross_attn_fewkeys: 1 <= P <= 8 keys per item");
    WIW_REQUIRE(rows > 0 && rows_per_item > 0 && rows % rows_per_item == 0, "cross_attn_fewkeys: rows must be whole items");
    const int C = heads * 64;
    WIW_REQUIRE(ldq % 8 == 0 && ldq >= C && ldo % 8 == 0 && ldo >= C, "cross_attn_fewkeys: ldq / ldo must be multiples of 8 and >= C");
    WIW_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) == 0, "cross_attn_fewkeys: pointers must be 16-byte aligned");
    const int items = (int)(rows / rows_per_item);
    int rpb = 256;                                   // rows per block: K / V (P x C x 2 x 2 bytes) are fetched once per block
    if (rpb > rows_per_item) rpb = rows_per_item;
    const int splits = (rows_per_item + rpb - 1) / rpb;
    hipLaunchKernelGGL(cross_attn_fewkeys_kernel, dim3((unsigned)splits, (unsigned)items), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)Q, ldq, (const uint16_t*)K, (const uint16_t*)V, (uint16_t*)O, ldo, C, P, rows_per_item, rpb,
                       scale * 1.4426950408889634f);
    return wiw_check_launch("wiw_cross_attn_fewkeys_bf16");
}
