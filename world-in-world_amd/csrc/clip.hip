// CLIP image-encoder front end on gfx950: the reference's `_encode_image`
// (dp/pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py:183-229) minus the linear layers, which run
// on wiw_gemm_bf16 / wiw_layernorm_bf16:
//   * wiw_clip_preprocess : `_resize_with_antialiasing` (pipeline:643-669: separable Gaussian blur with reflect padding,
//     then bicubic interpolation with align_corners=True, A = -0.75) + `(x + 1) / 2` + CLIP mean / std normalisation
//     (pipeline:199-208) + the im2col of the stride-P patch convolution, written straight into the bf16 operand rows of
//     the patch-embedding GEMM.  Blur and interpolation are both linear and separable, so each image axis is ONE pass:
//     out = (R_y B_y) (R_x B_x) img — the reference's order (B_x, B_y, then R) differs in fp32 rounding only.
//   * wiw_attn_small_bf16 : softmax(Q K^T * scale) V for short sequences and any head_dim that is a multiple of 16 up to
//     128 (ViT-H/14: 257 tokens, 16 heads of 80) — the UNet's attention kernels are specialised for head_dim 64.  One wave
//     per (sequence, head, 16 queries); same register formulation as attention.hip (S^T = K.Q^T, O^T = V^T.P^T: the
//     probabilities never leave the registers), K / Q / V^T fragments straight from global memory (the whole problem is
//     L2-resident), online softmax in fp32.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------
// resize: horizontal pass.  img fp32 [planes][H0][W0] in [-1,1] -> tmp fp32 [planes][H0][OUT]
// ---------------------------------------------------------------------------------------------
struct Taps {
    float k[32];   // Gaussian taps (<= 31)
    int n;         // number of taps (odd)
};

WIW_DEV int reflect_idx(int i, int n) {   // F.pad(mode="reflect"): -1 -> 1, n -> n - 2
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

WIW_DEV void cubic_weights(float t, float* w) {   // cubic convolution, A = -0.75 (upsample_bicubic2d)
    const float A = -0.75f;
    const float x0 = t + 1.0f, x3 = 2.0f - t, u = 1.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

__global__ void clip_resize_h_kernel(const float* __restrict__ img, int planes, int H0, int W0, int OUT, float scale,
                                     const Taps tx, float* __restrict__ tmp) {
    const int64_t total = (int64_t)planes * H0 * OUT;
    const int padl = (tx.n - 1) / 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(idx % OUT);
        const int64_t row = idx / OUT;                  // plane * H0 + y
        const float* src = img + row * W0;
        const float sx = scale * (float)xo;
        const int x0 = (int)floorf(sx);
        float w[4];
        cubic_weights(sx - (float)x0, w);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int u = x0 - 1 + j;
            u = u < 0 ? 0 : (u > W0 - 1 ? W0 - 1 : u);  // bicubic taps clamp to the image
            float b = 0.f;
            for (int k = 0; k < tx.n; ++k) b += tx.k[k] * src[reflect_idx(u + k - padl, W0)];
            acc += w[j] * b;
        }
        tmp[idx] = acc;
    }
}

// vertical pass + normalisation + im2col.  tmp fp32 [B*3][H0][OUT] -> A bf16 [B*rows_per_image][ldA]:
//   row  b*rows_per_image + 1 + (y/P)*(OUT/P) + x/P,   column  c*P*P + (y%P)*P + x%P     (conv weight flattened (c, ky, kx))
__global__ void clip_resize_v_patchify_kernel(const float* __restrict__ tmp, int B, int H0, int OUT, float scale,
                                              const Taps ty, float m0, float m1, float m2, float is0, float is1, float is2,
                                              int P, int rows_per_image, int ldA, uint16_t* __restrict__ A) {
    const int64_t total = (int64_t)B * 3 * OUT * OUT;
    const int padt = (ty.n - 1) / 2;
    const int grid = OUT / P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % OUT);
        const int y = (int)((idx / OUT) % OUT);
        const int c = (int)((idx / ((int64_t)OUT * OUT)) % 3);
        const int b = (int)(idx / ((int64_t)3 * OUT * OUT));
        const float* src = tmp + ((int64_t)(b * 3 + c) * H0) * OUT + x;
        const float sy = scale * (float)y;
        const int y0 = (int)floorf(sy);
        float w[4];
        cubic_weights(sy - (float)y0, w);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int u = y0 - 1 + i;
            u = u < 0 ? 0 : (u > H0 - 1 ? H0 - 1 : u);
            float bsum = 0.f;
            for (int k = 0; k < ty.n; ++k) bsum += ty.k[k] * src[(int64_t)reflect_idx(u + k - padt, H0) * OUT];
            acc += w[i] * bsum;
        }
        const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
        const float v = ((acc + 1.0f) * 0.5f - mean) * istd;
        const int64_t rowA = (int64_t)b * rows_per_image + 1 + (y / P) * grid + x / P;
        A[rowA * ldA + c * P * P + (y % P) * P + (x % P)] = f2bf(v);
    }
}

// ---------------------------------------------------------------------------------------------
// attention for short sequences, head_dim D (multiple of 16, <= 128)
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_small_kernel(const uint16_t* __restrict__ QK, int ldqk, int k_col_off,
                                                          const uint16_t* __restrict__ Vt, int64_t ldvt,
                                                          uint16_t* __restrict__ O, int ldo, int S, int Sp, int heads,
                                                          int q_tiles, int64_t total, float scale_log2e) {
    constexpr int NKK = (D + 31) / 32;   // 32-deep steps over d for S^T
    constexpr int NDB = D / 16;          // 16-row blocks of O^T
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= total) return;           // wave-uniform, no block barrier below
    const int qt = (int)(task % q_tiles);
    const int h = (int)((task / q_tiles) % heads);
    const int64_t seq = task / ((int64_t)q_tiles * heads);
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t row0 = seq * Sp;
    const bf16x8 zero8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

    bf16x8 qf[NKK];
    {
        int qi = qt * 16 + fr;
        qi = qi < Sp ? qi : Sp - 1;
        const uint16_t* src = QK + (row0 + qi) * ldqk + h * D;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = (kk * 32 + fq * 8 < D) ? *(const bf16x8*)(src + kk * 32 + fq * 8) : zero8;
    }
    f32x4 o[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int nkt = (S + 15) / 16;       // key tiles; Sp is a multiple of 16, so every tile stays inside the sequence's rows
    for (int kt = 0; kt < nkt; ++kt) {
        const uint16_t* ksrc = QK + (row0 + kt * 16 + fr) * ldqk + k_col_off + h * D;
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const bf16x8 kf = (kk * 32 + fq * 8 < D) ? *(const bf16x8*)(ksrc + kk * 32 + fq * 8) : zero8;
            st = WIW_MFMA(kf, qf[kk], st);   // lane: query fr, keys 4*fq + r
        }
        float sv[4], mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sv[r] = (kt * 16 + fq * 4 + r) < S ? st[r] * scale_log2e : -INFINITY;
            mx = fmaxf(mx, sv[r]);
        }
        mx = xor32_max(xor16_max(mx));
        const float m_new = fmaxf(m_run, mx);                 // finite from the first tile on (key 0 is always valid)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float p[4], ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(sv[r] - m_new); ls += p[r]; }
        ls = xor32_sum(xor16_sum(ls));
        l_run = l_run * alpha + ls;
        m_run = m_new;
        union { uint32_t u[4]; bf16x8 v; } po;
        po.u[0] = pack2bf(p[0], p[1]); po.u[1] = pack2bf(p[2], p[3]); po.u[2] = 0u; po.u[3] = 0u;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            // V^T fragment: lane = d row (db*16 + fr), 4 consecutive keys 16*kt + 4*fq .. +3 (8-byte aligned: Sp % 16 == 0)
            const uint2 vv = *(const uint2*)(Vt + (int64_t)(h * D + db * 16 + fr) * ldvt + row0 + kt * 16 + fq * 4);
            union { uint32_t u[4]; bf16x8 v; } vo;
            vo.u[0] = vv.x; vo.u[1] = vv.y; vo.u[2] = 0u; vo.u[3] = 0u;
            f32x4 acc = o[db];
            acc[0] *= alpha; acc[1] *= alpha; acc[2] *= alpha; acc[3] *= alpha;
            o[db] = WIW_MFMA(vo.v, po.v, acc);   // lane: query fr, d = 16*db + 4*fq + r
        }
    }
    const int qi = qt * 16 + fr;
    if (qi < Sp) {   // padding query rows (S <= qi < Sp) are written as ZEROS: downstream GEMMs / LayerNorms stay finite
        const float inv = qi < S ? 1.0f / l_run : 0.0f;
        uint16_t* dst = O + (row0 + qi) * ldo + h * D + fq * 4;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
            uint2 pk;
            pk.x = pack2bf(o[db][0] * inv, o[db][1] * inv);
            pk.y = pack2bf(o[db][2] * inv, o[db][3] * inv);
            if (qi >= S) pk = uint2{0u, 0u};
            *(uint2*)(dst + db * 16) = pk;
        }
    }
}

template <int D>
void launch_small(hipStream_t s, const uint16_t* QK, int ldqk, int k_col_off, const uint16_t* Vt, int64_t ldvt, uint16_t* O,
                  int ldo, int seqs, int S, int Sp, int heads, float scale) {
    const int q_tiles = Sp / 16;   // all rows of a sequence, padding included (written as zeros)
    const int64_t total = (int64_t)seqs * heads * q_tiles;
    hipLaunchKernelGGL((attn_small_kernel<D>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, QK, ldqk, k_col_off, Vt,
                       ldvt, O, ldo, S, Sp, heads, q_tiles, total, scale * LOG2E);
}

inline int grid_for(int64_t total, int block, int cap) {
    int64_t g = (total + block - 1) / block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int wiw_attn_small_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt,
                                   void* O, int ldo, int seqs, int S, int Sp, int heads, int head_dim, float scale) {
    WIW_REQUIRE(QK && Vt && O, "attn_small: null pointer");
    WIW_REQUIRE(seqs > 0 && S > 0 && heads > 0 && Sp >= S, "attn_small: bad sizes");
    WIW_REQUIRE(Sp % 16 == 0, "attn_small: the row stride of a sequence (Sp) must be a multiple of 16");
    WIW_REQUIRE(head_dim % 16 == 0 && head_dim >= 16 && head_dim <= 128, "attn_small: head_dim must be a multiple of 16, <= 128");
    WIW_REQUIRE(ldqk % 8 == 0 && k_col_off % 8 == 0 && ldvt % 4 == 0 && ldo % 4 == 0, "attn_small: misaligned strides");
    WIW_REQUIRE((int64_t)seqs * heads * (Sp / 16) < (1ll << 32), "attn_small: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)QK;
    const uint16_t* v = (const uint16_t*)Vt;
    uint16_t* o = (uint16_t*)O;
    switch (head_dim) {
        case 16: launch_small<16>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 32: launch_small<32>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 48: launch_small<48>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 64: launch_small<64>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 80: launch_small<80>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 96: launch_small<96>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        case 112: launch_small<112>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
        default: launch_small<128>(s, q, ldqk, k_col_off, v, ldvt, o, ldo, seqs, S, Sp, heads, scale); break;
    }
    return wiw_check_launch("wiw_attn_small_bf16");
}

extern "C" int wiw_clip_preprocess(void* stream, const float* img, int B, int H0, int W0, int out_size, int patch,
                                   const float* taps_x, int ntx, const float* taps_y, int nty, const float* mean,
                                   const float* inv_std, float* tmp, void* A, int rows_per_image, int ldA) {
    WIW_REQUIRE(img && taps_x && taps_y && mean && inv_std && tmp && A, "clip_preprocess: null pointer");
    WIW_REQUIRE(B > 0 && H0 > 1 && W0 > 1 && out_size > 1 && patch > 0 && out_size % patch == 0, "clip_preprocess: bad geometry");
    WIW_REQUIRE(ntx >= 1 && ntx <= 31 && (ntx & 1) && nty >= 1 && nty <= 31 && (nty & 1), "clip_preprocess: odd tap counts <= 31");
    WIW_REQUIRE((ntx - 1) / 2 < W0 && (nty - 1) / 2 < H0, "clip_preprocess: blur wider than the image");
    const int grid = out_size / patch;
    WIW_REQUIRE(rows_per_image >= 1 + grid * grid && ldA >= 3 * patch * patch, "clip_preprocess: operand buffer too small");
    Taps tx, ty;
    for (int i = 0; i < 32; ++i) { tx.k[i] = i < ntx ? taps_x[i] : 0.f; ty.k[i] = i < nty ? taps_y[i] : 0.f; }   // HOST arrays
    tx.n = ntx; ty.n = nty;
    hipStream_t s = (hipStream_t)stream;
    const float sx = (float)(W0 - 1) / (float)(out_size - 1), sy = (float)(H0 - 1) / (float)(out_size - 1);
    hipLaunchKernelGGL(clip_resize_h_kernel, dim3(grid_for((int64_t)B * 3 * H0 * out_size, 256, 4096)), dim3(256), 0, s, img,
                       B * 3, H0, W0, out_size, sx, tx, tmp);
    hipLaunchKernelGGL(clip_resize_v_patchify_kernel, dim3(grid_for((int64_t)B * 3 * out_size * out_size, 256, 4096)), dim3(256),
                       0, s, tmp, B, H0, out_size, sy, ty, mean[0], mean[1], mean[2], inv_std[0], inv_std[1], inv_std[2], patch,
                       rows_per_image, ldA, (uint16_t*)A);
    return wiw_check_launch("wiw_clip_preprocess");
}
