/*
 * wiw_svd.h — C ABI of the MI355X-native SVD denoising hot path (libwiwsvd.so).
 *
 * The reference (World-In-World) has no FFI: its hot path is Python calling torch.nn modules
 * (SURVEY.md §8b).  This header is therefore the boundary DESIGNED by the build: one entry point per
 * device operator that the reference executes implicitly through ATen/cuDNN/cuBLAS/SDPA.  Each
 * declaration cites the reference call site(s) it replaces; `dp/` =
 * FTsvd/diffusers-private/diffusers/.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); no internal threads,
 *     no allocation, no synchronisation inside any entry point (graph-capture safe);
 *   - activations are bf16 (uint16 storage) token-major: X[m][c], m = ((b*T + t)*H + y)*W + x
 *     (NHWC per frame); weights bf16 [out][k]; conv weights bf16 [Cout][tap][Cin];
 *     vectors (bias, gamma, beta, statistics, embeddings, latents) fp32;
 *   - return value: WIW_OK or a negative WIW_E* code (nothing is launched on error);
 *     wiw_last_error() returns a static message for the calling thread's last failure.
 */
#ifndef WIW_SVD_H
#define WIW_SVD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WIW_OK 0
#define WIW_EINVAL -1  /* unsupported shape / null pointer / misaligned argument */
#define WIW_ELAUNCH -2 /* HIP launch error */
#define WIW_ENODEV -3  /* no gfx950 device visible */

#define WIW_ABI_VERSION 17  /* 2: WiwGemmArgs gained A3 / C3 (fused conv3x3 + shortcut segment);
                             3: wiw_temporal_attn_block_bf16; GroupNorm stats = (mean, variance);
                             4: WiwGemmArgs gained splitk / workspace; wiw_transpose_bf16;
                             5: wiw_dtype (the library exists in a bf16 and an fp16 build);
                             6: WiwGemmArgs gained lnfold / ln_eps (WIW_EPI_LNFOLD);
                             7: the training entry points settled: wiw_colsum sums CONTIGUOUS row ranges, wiw_attn_bwd_bf16
                                takes NULL transposes on its LDS-tiled path, wiw_gather_taps_t_bf16, wiw_wgrad_tn_bf16;
                             8: wiw_ffn_geglu_bf16 (fused LayerNorm + GEGLU FeedForward of the C = 320 level);
                             9: wiw_ema_step_f32 (EMAModel.step of --use_ema);
                             10: wiw_attn_spatial_lse_bf16 / wiw_attn_bwd_given_lse_bf16 (the training forward hands its row
                                 log-sum-exp to the backward);
                             11: the fp32 residual stream: WIW_EPI_RES1_F32 / WIW_EPI_RES2_F32, WIW_EPI_OUT_F32 on vectorised
                                 stores, wiw_groupnorm_stats_f32in / wiw_groupnorm_apply_stats_f32in / wiw_layernorm_f32in /
                                 wiw_cast_f32_to_16, wiw_calib_mfma;
                             12: wiw_groupnorm_stats / _f32in take `counters`: the second reduction stage runs inside the
                                 statistics launch (wiw_groupnorm_counters); WIW_K_HALO32 / wiw_conv_halo_ok; wiw_ffn_geglu_f32stream;
                             13: wiw_attn_spatial_ps_bf16 (32x32x16 spatial attention on a pre-scaled Q);
                             14: wiw_ffn32_geglu (the fused FeedForward on 32x32x16 MFMAs, weights in the sw16 tiling)
                                15: wiw_groupnorm_onepass / wiw_groupnorm_onepass_ok (one-pass GroupNorm of the inner levels);
                             16: wiw_ffn_geglu_f32stream2 (the fused FeedForward's LayerNorm reads the fp32 stream; a second,
                                 16-bit output); wiw_cross_attn_fewkeys_bf16 (cross-attention over 2..8 conditioning tokens);
                             17: wiw_calib_mfma_random (the box calibration loop on random operands: the SUSTAINED matrix-pipe rate) */

int wiw_abi_version(void);

/* 16-bit storage type of THIS library: 0 = bfloat16 (libwiwsvd.so), 1 = IEEE half (libwiwsvd_f16.so, the same sources
 * compiled with -DWIW_F16: the reference serves fp16 by default, FTsvd/eval_inference.py:294).  Every "bf16" in the
 * entry-point names and comments below reads "the library's 16-bit type": tensors are 16-bit words of that type,
 * accumulation / statistics / softmax are fp32 in both builds. */
int wiw_dtype(void);
const char* wiw_last_error(void);
/* Returns WIW_OK when device `dev` is a gfx950 (MI355X) and fills optional name buffer. */
int wiw_device_check(int dev, char* name, int name_len);

/* ------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution on bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate.
 *
 *   acc[m][n]  = sum_k Agather[m][k] * W[n][k]
 *   y          = alpha * (acc + bias[n] + rowvec[m / rows_per_vec][n]) (optionally SiLU / GELU / quick-GELU)
 *   out[m][n]  = y + beta1 * res1[m][n] + beta2 * res2[m][n]
 *   GEGLU      : W rows are stored in tiles of 160 = [80 value | 80 gate] (see weights.py
 *                `pack_geglu`); out[m][j] = (v + bias_v) * gelu_erf(g + bias_g), N counts packed rows.
 *
 * mode WIW_A_DENSE     : Agather[m][k] = concat(A[m][0:C1], A2[m][0:C2])           (K = C1 + C2)
 *        replaces nn.Linear / 1x1 conv call sites: dp/models/attention_processor.py:2358-2391
 *        (to_q/k/v/out), dp/models/attention.py:1185-1243 + activations.py:117-123 (GEGLU FF),
 *        dp/models/transformers/transformer_temporal.py:327,374 (proj_in/out),
 *        dp/models/resnet.py:311-318 (conv_shortcut over the skip concat, unet_3d_blocks.py:1612),
 *        dp/models/embeddings.py:804-816 (TimestepEmbedding), resnet.py:343-350 (time_emb_proj).
 * mode WIW_A_CONV3X3   : 3x3, stride 1, zero pad 1 over (H, Wd)                   (K = 9 * C1)
 *        dp/models/resnet.py:269,285 (ResnetBlock2D.conv1/conv2), unet:130-135, 255-260.
 *        With C2 > 0 a tenth, dense K segment concat(A2[m][0:C2], A3[m][0:C3]) follows the nine taps
 *        (K = 9 * C1 + C2 + C3): ResnetBlock2D's 1x1 `conv_shortcut` over the skip concat (resnet.py:311-318,
 *        unet_3d_blocks.py:1612) accumulated inside conv2 — the shortcut tensor is never written.
 * mode WIW_A_CONV3X3_S2: 3x3, stride 2, pad 1; input is (2H, 2Wd)                 (K = 9 * C1)
 *        dp/models/downsampling.py:132-150 (Downsample2D).
 * mode WIW_A_CONV3X3_S2P: 3x3, stride 2, zero pad (0,1,0,1) = bottom / right only; input is (2H, 2Wd)
 *        dp/models/downsampling.py:132-150 with padding=0 (VAE encoder DownEncoderBlock2D).
 * mode WIW_A_CONV3X3_UP: nearest x2 upsample fused into a 3x3 conv; input (H/2, Wd/2)
 *        dp/models/upsampling.py:142-186 (Upsample2D).
 * mode WIW_A_CONV_T3   : (3,1,1) temporal conv, zero pad 1 over T; S = H*Wd       (K = 3 * C1)
 *        dp/models/resnet.py:570-592 (TemporalResnetBlock.conv1/conv2).
 * W layouts: plain bf16 [N][K], or (epilogue bit WIW_W_TILED, static weights) TILED: ceil(N/8) x (K/64) blocks of 1 KiB,
 *   block (nb, kt) = rows 8*nb .. 8*nb+7 (zero rows past N), k-values 64*kt .. 64*kt+63; inside a block row r occupies bytes
 *   r*128 .. r*128+127 and its eight 16-byte chunks are stored XOR-swizzled: position p holds chunk p ^ r.  One LDS-DMA
 *   instruction of the kernel then reads one contiguous KiB — 63 B/clk/CU against 25 B/clk/CU for eight row segments K*2 bytes
 *   apart (tools/ubench/lds_fill.hip); the LDS image and every result bit are identical.  (`hip.tile_weight` builds it.)
 * LayerNorm fold (epilogue bit WIW_EPI_LNFOLD, dense mode, C2 == 0): A is the RAW input x of an nn.LayerNorm over K = C1
 *   (dp/models/attention.py:659-694 norm1 / norm3 / norm_in feeding to_q|k|v and the GEGLU projections) and W holds
 *   W * gamma (rounded to the 16-bit type):  LayerNorm(x) . W_n = rstd * (x . W'_n - mean * s_n) + t_n  with
 *   lnfold = [s (N) | t (N)], s_n = sum_k W'[n][k] (of the ROUNDED rows), t_n = sum_k W[n][k] * beta[k] + bias_n (bias must be
 *   NULL: it is inside t).  Row mean / rstd are accumulated inside the kernel from the operand fragments; the fold is
 *   applied to the fp32 accumulators before the first rounding.  N % 160 == 0, alpha == 1, 16-bit aligned output only;
 *   with WIW_EPI_GEGLU, s and t follow the packed row order of W.  Served by the 256x160 / 128x160 tiles.
 * Constraints: C1, C2, C3 % 64 == 0; K = C1 + C2 (dense), taps * C1 (conv) or 9 * C1 + C2 + C3 (conv3x3 + shortcut);
 * A2 only with WIW_A_DENSE / WIW_A_CONV3X3, A3 only with WIW_A_CONV3X3.
 * ---------------------------------------------------------------------------------------------- */
enum { WIW_A_DENSE = 0, WIW_A_CONV3X3 = 1, WIW_A_CONV3X3_S2 = 2, WIW_A_CONV3X3_UP = 3, WIW_A_CONV_T3 = 4,
       WIW_A_CONV3X3_S2P = 5 };
enum { WIW_EPI_GEGLU = 1, WIW_EPI_SILU = 2, WIW_EPI_OUT_F32 = 4,
       WIW_EPI_GELU = 8,        /* y = gelu_erf(y)           (CLIP ViT-H MLP, transformers `gelu`) */
       WIW_EPI_QUICK_GELU = 16, /* y = y * sigmoid(1.702 y)  (OpenAI CLIP `quick_gelu`) */
       WIW_W_TILED = 32,        /* W is pre-tiled for the LDS-DMA stream (above) */
       WIW_EPI_LNFOLD = 64,     /* A is the un-normalised LayerNorm input, W = W * gamma, lnfold = [s | t] (above) */
       /* The fp32 RESIDUAL STREAM (ABI 11).  The reference keeps its latents and Euler update in fp32
        * (scheduling_euler_discrete.py:635,673) and evaluates the UNet in fp16 (eval_inference.py:294); a 16-bit
        * library that also ROUNDS the residual stream x + f(x) after every block is 1.2e-3 (fp16) from the reference's
        * fp32 output on the served architecture, with the stream kept in fp32 it is below north_star's 1e-3
        * (oracle/precision_study.py).  res1 / res2 are then fp32 [M][ldr] and, with WIW_EPI_OUT_F32, so is
        * `out`; MFMA operands stay 16-bit.  Launches with any of these three bits run the 256x160 / 128x160 tiles with
        * a fragment-layout epilogue: fp32 accumulator + bias + vector + residuals, ONE rounding (none for fp32 out),
        * 16-byte accesses (N, ldo, ldr1, ldr2 % 4 == 0; no GEGLU / SiLU / GELU). */
       WIW_EPI_RES1_F32 = 128, WIW_EPI_RES2_F32 = 256,
       /* Convolution modes, ABI 11: the K index of W (and of the kernel's tap walk) is CHANNEL-BLOCK major,
        *     k = ((c / 64) * taps + tap) * 64 + c % 64        instead of        k = tap * C1 + c,
        * so the `taps` K tiles that read (shifted windows of) the SAME 64 channels of the activation are consecutive: the
        * window stays in the XCD's L2 between them.  In tap-major order every tap streams the whole activation panel
        * through the 4-MiB L2 before the next tap touches it again, and the PMC FETCH_SIZE of the 3x3 convolutions was 9x
        * their algorithmic reads (1.39 GB per launch at 2.4 TB/s).  The fused shortcut segment (C2 + C3) stays behind the
        * taps.  Pure permutation of K: the host re-orders W (`unet.py`), results differ only by fp32 summation order. */
       WIW_K_CMAJOR = 512,
       /* WIW_A_CONV3X3 and WIW_A_CONV3X3_UP, ABI 12: HALO-STAGED A operand.  K index of W (and of the kernel's walk) in 32-channel blocks,
        *     k = ((c / 32) * 9 + tap) * 32 + c % 32,
        * and the kernel (256x320 tile) stages, per 32-channel block, the (R + 2) x (Wd + 2) pixel neighbourhood of its
        * R = 256 / Wd image rows in LDS ONCE; the nine taps are nine shifted reads of that image instead of nine LDS-DMA
        * fetches of a [256][64]-element tile (1/9 of the A bytes through the DMA + the halo).  Geometry the kernel takes
        * (wiw_conv_halo_ok; anything else is refused — W's K order belongs to this kernel, the caller keeps a second copy
        * of W in one of the other orders for other geometries): Wd = 32, 64 or 128, H >= 256 / Wd, M % 256 == 0 (a tile
        * that straddles two frames gets a row of zeros between them), N % 320 == 0, W tiled (WIW_W_TILED), 16-bit output
        * through the staged epilogue, no split-K.  The fused shortcut segment (A2 / A3, K = 9 * C1 + C2 + C3) is walked
        * after the taps in plain channel order, as in the other K orders.  WIW_A_CONV3X3_UP: the staged image is the
        * low-resolution input (R / 2 + 2 rows of Wd / 2 + 2 pixels); Wd = 64 or 128, H even, (H * Wd) % 256 == 0, no segment. */
       WIW_K_HALO32 = 1024 };

typedef struct WiwGemmArgs {
    const void* A;       /* bf16 [rows_in][C1] */
    const void* A2;      /* bf16 [rows_in][C2] or NULL */
    const void* W;       /* bf16 [N][K] */
    void* out;           /* bf16 (or fp32 with WIW_EPI_OUT_F32) [M][ldo] */
    const float* bias;   /* [N] or NULL */
    const float* rowvec; /* [M / rows_per_vec][rowvec_ld] or NULL */
    const void* res1;    /* bf16 [M][ldr1] (fp32 with WIW_EPI_RES1_F32) or NULL */
    const void* res2;    /* bf16 [M][ldr2] (fp32 with WIW_EPI_RES2_F32) or NULL */
    const void* zeros;   /* >= 16 bytes of device zeros (source of padding taps) */
    const void* A3;      /* bf16 [rows_in][C3] or NULL (second half of the fused shortcut segment, CONV3X3 only) */
    int32_t M, N, K;
    int32_t C1, C2, C3;
    int32_t mode;
    int32_t H, Wd, T;    /* OUTPUT geometry of the conv modes */
    int32_t ldo, ldr1, ldr2;
    int32_t n_out;       /* GEGLU only: number of valid output columns (N/2 minus packing pad) */
    int32_t rowvec_ld, rows_per_vec;
    float alpha, beta1, beta2;
    int32_t epilogue;    /* WIW_EPI_* bits */
    int32_t splitk;      /* 0 / 1: off.  S > 1: the K loop is cut into S equal ranges (K / 64 divisible by S) whose raw fp32
                          * sums go to `workspace`; a second kernel adds them IN RANGE ORDER and applies the epilogue
                          * (deterministic; for launches whose M alone cannot fill 256 CUs).  Not with GEGLU / GELU. */
    void* workspace;     /* split-K only: S * M * N floats of device scratch */
    const float* lnfold; /* WIW_EPI_LNFOLD only: [s (N) | t (N)] fp32, see below */
    float ln_eps;        /* WIW_EPI_LNFOLD only: the LayerNorm epsilon */
} WiwGemmArgs;

int wiw_gemm_bf16(void* stream, const WiwGemmArgs* args);
/* 1 when `args` (with WIW_K_HALO32 set) is a launch the halo-staged convolution kernel takes, else 0 (no launch). */
int wiw_conv_halo_ok(const WiwGemmArgs* args);

/* ------------------------------------------------------------------------------------------------
 * Spatial self-attention, head_dim 64, flash-style (online softmax, fp32 statistics):
 *   O[n][s][h*64 + d] = softmax_s'(Q.K / 8) V,  per frame n and head h.
 * Replaces F.scaled_dot_product_attention at dp/models/attention_processor.py:2383-2385 as called
 * from BasicTransformerBlock.attn1 (dp/models/attention.py:507-512).
 *   QK : bf16 [frames*S][ldqk], Q at column h*64, K at column k_col_off + h*64
 *   Vt : bf16 [heads*64][ldvt] (V TRANSPOSED: Vt[h*64 + d][n*S + s]), produced by wiw_transpose_bf16 from the V
 *        columns of one fused q|k|v projection (or by wiw_gemm_bf16 with swapped operands); S % 8 == 0 required
 *        (16-byte aligned key runs).
 *   O  : bf16 [frames*S][ldo];  zeros: >= 16 bytes of device zeros (V^T chunks past the frame end)
 * ---------------------------------------------------------------------------------------------- */
int wiw_attn_spatial_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt,
                          void* O, int ldo, int frames, int S, int heads, float scale, const void* zeros);

/* The same operator on a PRE-SCALED Q (ABI 13): the caller folded log2(e) / sqrt(64) into the to_q rows of the q|k|v
 * projection before their rounding to 16 bits (dp/models/attention_processor.py:2358-2366 computes the three projections
 * separately and F.scaled_dot_product_attention, :2383, applies 1/sqrt(d)), so that Q.K is the base-2 exponent itself.
 * S % 128 == 0, S >= 256 (the 72x128, 36x64 and 32x32 levels): v_mfma_f32_32x32x16 kernel, the softmax of one 32-key half
 * tile issued between the matrix instructions of the next (csrc/attention32.hip); other S: the kernel above with scale = 1.
 * Same operand layouts as wiw_attn_spatial_bf16; ldo % 8 == 0. */
int wiw_attn_spatial_ps_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt,
                             void* O, int ldo, int frames, int S, int heads, const void* zeros);

/* ------------------------------------------------------------------------------------------------
 * Temporal self-attention over the T (<= 16) frames of every spatial site, head_dim 64:
 * rows of site (b, s) are m = (b*T + t)*S + s — the (B*S, T, C) permutes of
 * dp/models/attention.py:720-722, 758-760 are never materialised.
 * Replaces TemporalBasicTransformerBlock.attn1 (dp/models/attention.py:735-737).
 *   QKV : bf16 [batch*T*S][ldqkv] with Q | K | V at columns 0, C, 2C (C = heads*64);  O: [..][ldo]
 * ---------------------------------------------------------------------------------------------- */
int wiw_attn_temporal_bf16(void* stream, const void* QKV, int ldqkv, void* O, int ldo, int batch, int T, int S,
                           int heads, float scale);

/* ------------------------------------------------------------------------------------------------
 * FUSED temporal attention block: LayerNorm -> Q/K/V projection -> softmax(Q K^T) V over the T (<= 14) frames of every
 * spatial site, ONE kernel; the 3C-wide QKV tensor of wiw_attn_temporal_bf16 never exists.  Replaces
 * TemporalBasicTransformerBlock.norm1 + attn1 up to (not including) its out-projection
 * (dp/models/attention.py:735-737; to_q/to_k/to_v of dp/models/attention_processor.py:2358-2366 — no bias;
 * nn.LayerNorm of attention.py:659-694), rows m = (b*T + t)*S + s as above.
 *   X    : bf16 [batch*T*S][C]  the residual stream BEFORE norm1 (C = heads*64)
 *   Wqkv : bf16 [heads*192][C], TILED as described for wiw_gemm_bf16 (WIW_W_TILED);  rows of head h = [to_q rows h*64.. | to_k rows | to_v rows], each row multiplied by the
 *          LayerNorm weight gamma (W' = W * gamma, rounded to bf16): the kernel runs its MFMAs on the raw rows of X
 *   fold : fp32 [heads][512]    per head: s[192] = sum_k W'[n][k] (of the bf16-rounded W'), t[192] = sum_k W[n][k]*beta[k],
 *          128 floats of padding;  q_n = rstd * (x . W'_n - mean * s_n) + t_n  (LayerNorm folded exactly; mean / rstd
 *          of each row are accumulated inside the kernel from the operand fragments, eps as given)
 *   O    : bf16 [batch*T*S][ldo], columns h*64 + d.   1 <= T <= 14; all pointers 16-byte aligned; ldo % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
int wiw_temporal_attn_block_bf16(void* stream, const void* X, const void* Wqkv, const float* fold, void* O, int ldo,
                                 int batch, int T, int S, int heads, float eps, float scale, const void* zeros);

/* ------------------------------------------------------------------------------------------------
 * FUSED FeedForward (GEGLU) of the 320-channel level, optionally with the LayerNorm in front of it, ONE kernel:
 *
 *   h   = GEGLU( LN?(X) . W1^T + b1 )            [M][1280], never written to memory
 *   out = alpha * ( h . W2^T + b2 + rowvec[m / rows_per_vec] ) + beta1 * res1 + beta2 * res2
 *
 * Replaces FeedForward.forward (dp/models/attention.py:1185-1243: net.0 = GEGLU proj, activations.py:93-123, exact-erf
 * GELU; net.2 = Linear) together with the residual add / AlphaBlender that follows it
 * (attention.py:565-582, 756-762; transformer_temporal.py:364-372), and with ln != 0 also norm3 / norm_in
 * (nn.LayerNorm, attention.py:540-567, 745-756).  Per 128-row tile the hidden activation lives in LDS / registers only.
 *   X    : bf16 [M][ldx]; ln == 0: the LayerNorm OUTPUT; ln != 0: the RAW LayerNorm input (statistics over the 320 columns,
 *          two-pass in registers, eps = ln_eps) — then W1 must hold W1 * gamma and b1 must hold W1 . beta + b1
 *   W1   : bf16 [2560][320], rows packed in chunks of 128 = [64 value rows | 64 gate rows] of hidden units 64c .. 64c+63
 *          (value rows = first half of net.0.proj.weight, gate rows = second half), TILED (WIW_W_TILED layout, above)
 *   b1   : fp32 [2560] packed the same way;   W2: bf16 [320][1280] TILED;   b2: fp32 [320] or NULL
 *   rowvec / rows_per_vec / res1 / res2 / alpha / beta1 / beta2 : as for wiw_gemm_bf16 (fp32 vector rows; bf16 residuals)
 *   C_in, hidden : must be 320, 1280 (anything else returns WIW_EINVAL: the wider levels run on wiw_gemm_bf16)
 * Rounding contract: as in the staged epilogue of wiw_gemm_bf16, alpha * (h . W2^T) is rounded to the 16-bit type when it is
 * staged for the row-major store phase, b2 / rowvec / residuals are then added in fp32 and the sum is rounded again on store —
 * one more 16-bit rounding of the FeedForward branch than a single fp32 epilogue would give (2^-9 resp. 2^-12 relative to the
 * BRANCH, not to the sum).  The parity gates of this kernel are against the fp32 oracle (tests/test_hip_ffn.py), and the
 * fp32-residual-stream mode does not use it (its FeedForward adds run in the one-rounding epilogue of wiw_gemm_bf16, ABI 11).
 * All pointers 16-byte aligned; ldx, ldo, ldr1, ldr2 multiples of 8; rowvec_ld a multiple of 4.
 * ---------------------------------------------------------------------------------------------- */
int wiw_ffn_geglu_bf16(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                       const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1, int ldr1,
                       float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out, int ldo, int64_t M,
                       int C_in, int hidden, int ln, float ln_eps);
/* ABI 12: the same operator on the fp32 RESIDUAL STREAM (ABI 11): f32 bit 0 = `out` is fp32 [M][ldo], bit 1 = res1 is
 * fp32, bit 2 = res2 is fp32 (leading dimensions in elements).  Bias, per-frame vector and residuals are added to the fp32
 * accumulators in the fragment layout and the sum is rounded once (not at all for an fp32 `out`).  The in-kernel LayerNorm
 * (`ln`) still reads a 16-bit X. */
int wiw_ffn_geglu_f32stream(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                       const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1, int ldr1,
                       float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out, int ldo, int64_t M,
                       int C_in, int hidden, int ln, float ln_eps, int f32);
/* ABI 16: the same operator with the fp32 stream on the INPUT side too.  f32 bit 3 = X is fp32 [M][ldx] (ldx in elements):
 * allowed with ln != 0 only — the in-kernel LayerNorm (attention.py:540-567 / 745-756) takes exact two-pass moments of the fp32
 * row and packs the normalised 16-bit MFMA operand, so the standalone wiw_layernorm_f32in pass in front of the FeedForward
 * (read 4 B + write 2 B per element) disappears.  out16 (may be NULL): a second output [M][ldo16] in the library's 16-bit
 * type = the rounding of `out` — what a 16-bit consumer of the stream (wiw_temporal_attn_block_bf16) reads, written from the
 * registers that hold the fp32 result instead of by a wiw_cast_f32_to_16 pass. */
int wiw_ffn_geglu_f32stream2(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                        const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1, int ldr1,
                        float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out, int ldo, int64_t M,
                        int C_in, int hidden, int ln, float ln_eps, int f32, void* out16, int ldo16);
/* ABI 14: the same operator, same arguments and f32 mask, on v_mfma_f32_32x32x16 with every tensor of a 128-row tile in
 * registers (csrc/ffn32.hip: one wave per SIMD, the exact-erf GEGLU of hidden chunk c - 1 issued between the MFMAs of chunk c;
 * no LDS staging in the epilogue: ONE rounding of alpha * (h . W2^T + b2 + rowvec) + beta1 res1 + beta2 res2 whatever the
 * output type).  Operand packing differs (`unet.pack_ffn32`):
 *   W1 : [2560][320] in chunks of 64 rows = [32 value | 32 gate] rows of hidden units 32c .. 32c+31, TILED with the 16-row
 *        chunk swizzle (position p of row R holds 16-byte chunk p ^ ((R >> 1) & 7); the default tiling uses p ^ (R & 7))
 *   b1 : fp32 [2560] packed the same way, its VALUE half pre-multiplied by 0.5
 *   W2 : [320][1280] TILED with the 16-row swizzle */
int wiw_ffn32_geglu(void* stream, const void* X, int ldx, const void* W1, const float* b1, const void* W2,
                    const float* b2, const float* rowvec, int rowvec_ld, int rows_per_vec, const void* res1, int ldr1,
                    float beta1, const void* res2, int ldr2, float beta2, float alpha, void* out, int ldo, int64_t M,
                    int C_in, int hidden, int ln, float ln_eps, int f32);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) in NHWC, split into statistics + fused normalise/affine/SiLU.
 * Replaces torch.nn.GroupNorm (+ SiLU) at dp/models/resnet.py:327-328, 351-363, 611-622,
 * dp/models/transformers/transformer_temporal.py:324, unet:565-566.  `rows_per_unit` = H*W for the
 * per-frame 2D norms and T*H*W for the 5D norms of TemporalResnetBlock (statistics over T too).
 * The input may be the channel concat of two tensors (skip connections, unet_3d_blocks.py:1612).
 *   stats : fp32 [units][32][2] (mean, biased variance) per group, fully overwritten.  Numerically stable: shifted
 *           sums per thread, then Chan merges of (count, mean, M2) — no E[x^2] - mean^2 cancellation (ABI v3;
 *           v2 stored raw (sum, sum of squares));
 *   scratch : fp32, >= wiw_groupnorm_scratch_floats(rows, rows_per_unit, rows_per_block) elements (per-block
 *             (mean, M2) pairs, then per-part (count, mean, M2) triples).  The reduction is deterministic: every merge
 *             order is fixed and no DATA goes through atomics.  ONE launch since ABI 12: a block publishes its pair behind
 *             an agent-scope release fence and bumps a counter; the block that finds its part of 32 blocks complete merges
 *             the part, the block that finds the unit's parts complete merges those — which block that is depends on
 *             timing, what it computes does not (before: a second launch, 8.7 us x 105 GroupNorms per UNet forward).
 *             rows_per_block (0 = default by unit size) fixes the summation order of a unit
 *             independently of how many units the call covers — a candidate's bits do not depend on its batch;
 *   counters : >= wiw_groupnorm_counters(rows, rows_per_unit, rows_per_block) unsigned, ZERO before the first call; every
 *              call leaves them zero.  Calls that may run concurrently (different streams) need different counters;
 *   ab    : fp32 [units][2][C]  per-channel scale a = rstd*gamma and shift b = beta - mean*a.
 * ---------------------------------------------------------------------------------------------- */
int64_t wiw_groupnorm_scratch_floats(int64_t rows, int rows_per_unit, int rows_per_block);
int64_t wiw_groupnorm_counters(int64_t rows, int rows_per_unit, int rows_per_block);
int wiw_groupnorm_stats(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                        int rows_per_unit, int rows_per_block, float* stats, float* scratch, unsigned* counters);
int wiw_groupnorm_finalize(void* stream, const float* stats, const float* gamma, const float* beta, int units,
                           int C, int rows_per_unit, float eps, float* ab);
int wiw_groupnorm_apply(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                        int rows_per_unit, const float* ab, int silu, void* out);
/* finalize + apply in one launch: scale / shift are derived from the raw `stats` inside the kernel. */
int wiw_groupnorm_apply_stats(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows,
                              int rows_per_unit, const float* stats, const float* gamma, const float* beta, float eps,
                              int silu, void* out);
/* ABI 11, fp32 residual stream: the same two passes over fp32 inputs X1 / X2 (fp32 [rows][C1], [rows][C2]); the
 * normalised output stays 16-bit.  `raw16` (NULL or 16-bit [rows][C1 + C2]) additionally receives the ROUNDED raw input
 * (concatenated): the MFMA operand of ResnetBlock2D's 1x1 conv_shortcut (resnet.py:311-318), which reads the block input
 * itself — one extra write in a pass that holds the value in registers anyway, instead of a cast pass. */
int wiw_groupnorm_stats_f32in(void* stream, const float* X1, int C1, const float* X2, int C2, int64_t rows,
                              int rows_per_unit, int rows_per_block, float* stats, float* scratch, unsigned* counters);
int wiw_groupnorm_apply_stats_f32in(void* stream, const float* X1, int C1, const float* X2, int C2, int64_t rows,
                                    int rows_per_unit, const float* stats, const float* gamma, const float* beta, float eps,
                                    int silu, void* out, void* raw16);

/* ABI 15 (round 5): GroupNorm(32) (+SiLU) of the two inner levels in ONE pass — a block keeps 160 consecutive channels of
 * one frame in registers (read once; exact two-pass statistics in fp32; normalise; store): the same operator as
 * wiw_groupnorm_stats + wiw_groupnorm_apply_stats (dp/models/resnet.py:262-300 `norm1` / `norm2`, transformer_temporal.py:
 * 250 `self.norm`), for rows_per_unit (= H * W of a frame) in {144, 576}, 40 or 80 channels per group, C1 and C2 multiples
 * of 160, 16-bit tensors.  wiw_groupnorm_onepass_ok(C1, C2, rows, rows_per_unit) != 0 says a launch fits; anything else is
 * refused (WIW_EINVAL), never re-routed.  Deterministic, no scratch buffers. */
int wiw_groupnorm_onepass_ok(int C1, int C2, int64_t rows, int rows_per_unit);
int wiw_groupnorm_onepass(void* stream, const void* X1, int C1, const void* X2, int C2, int64_t rows, int rows_per_unit,
                          const float* gamma, const float* beta, float eps, int silu, void* out);

/* ------------------------------------------------------------------------------------------------
 * ABI 16: cross-attention over a FEW keys — BasicTransformerBlock.attn2 (dp/models/attention.py:545-551) and
 * TemporalBasicTransformerBlock.attn2 (:740-743) through AttnProcessor2_0 (attention_processor.py:2358-2391) when the
 * conditioning holds P > 1 tokens per candidate (--num_past_obs > 1: train_svd.py:359, 889-894; pipeline_stable_video_
 * diffusion.py:500-508).  Replaces F.scaled_dot_product_attention(q, k, v) with Sk = P:
 *   O[m][h*64+d] = sum_p softmax_p(Q[m][h*64:(h+1)*64] . K[m / rows_per_item][p][h*64:(h+1)*64] * scale) V[...][p][h*64+d]
 * Q [rows][ldq], O [rows][ldo]: 16-bit token-major tensors (heads * 64 columns used); K, V: 16-bit [rows / rows_per_item][P]
 * [heads*64] contiguous (the projections to_k / to_v of the item's P embeddings); 1 <= P <= 8, heads * 64 <= 2048; fp32
 * softmax.  The rows of an item (T * S in this build's token order, for the spatial and the temporal block alike) share
 * its keys.  With P = 1 the served path never calls this: the operator collapses to a vector per candidate (unet.py). */
int wiw_cross_attn_fewkeys_bf16(void* stream, const void* Q, int ldq, const void* K, const void* V, void* O, int ldo,
                                int64_t rows, int rows_per_item, int heads, int P, float scale);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channel dim with an optional fused pre-add of a per-row-group vector:
 *   x' = x + addvec[m / rows_per_vec][:]   (written to `sum_out` when non-NULL)
 *   y  = LayerNorm(x') * gamma + beta      (eps, fp32 two-pass statistics, wave shuffles)
 * Replaces nn.LayerNorm at dp/models/attention.py:371,401,432,659-694, the `hidden_states + emb`
 * of transformer_temporal.py:352-353 and the residual add of the single-key cross-attentions
 * (attention.py:545-551, 740-743; exact elision).   C % 8 == 0, C <= 2048.
 * ---------------------------------------------------------------------------------------------- */
int wiw_layernorm_bf16(void* stream, const void* X, int64_t rows, int C, const float* gamma, const float* beta,
                       float eps, const float* addvec, int addvec_ld, int rows_per_vec, void* sum_out, void* out);
/* ABI 11, fp32 residual stream: X fp32 [rows][C], out 16-bit (no pre-add: the vectors ride in the producing GEMM). */
int wiw_layernorm_f32in(void* stream, const float* X, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                        void* out);
/* ABI 11: out[i] = round16(X[i]), n % 8 == 0 — the 16-bit MFMA operand of a stream tensor that enters a convolution
 * directly (Downsample2D / Upsample2D inputs, downsampling.py:132-150, upsampling.py:142-186). */
int wiw_cast_f32_to_16(void* stream, const float* X, int64_t n, void* out);
/* ABI 11, measurement only (bench.py `box`): `blocks` workgroups of 8 waves issue `iters` x 8 back-to-back MFMAs
 * (v_mfma_f32_16x16x32, register operands, no memory traffic) per wave: blocks * 8 * iters * 8 * 16384 flop.  Timed by the
 * caller; what this box's clocks give the matrix pipe (the pool's boxes differ by +-5 %). */
int wiw_calib_mfma(void* stream, int blocks, int iters, float* out);
/* ABI 17: the same loop on random operands — src: 128 x 16 bytes of the library's 16-bit type (lane l: A = entry l, B = entry
 * 64 + l), 16-byte aligned.  Held for about a second it measures what the board's power management lets the matrix pipe sustain
 * (bench.py `box.mfma_sustained`); same flop count as wiw_calib_mfma. */
int wiw_calib_mfma_random(void* stream, int blocks, int iters, const void* src, float* out);

/* ------------------------------------------------------------------------------------------------
 * Conditioning embedding rows (unet:464-487, micro_cond, no-grad path) with the SiLU of
 * resnet.py:343-344 applied once:  out[(i*T + t)][c] = silu(time[i][c] + act[(i % B)*T + t][c] + noise[i][c])
 * (contract of SURVEY.md §9.2: candidate i % B's actions — NOT the reference's cross-wiring).
 *   time, noise: fp32 [Bc][E];  act: fp32 [B*T][E];  out: bf16 [Bc*T][E]
 * ---------------------------------------------------------------------------------------------- */
int wiw_emb_combine(void* stream, const float* time, const float* act, const float* noise, int Bc, int B, int T,
                    int E, void* out);

/* ------------------------------------------------------------------------------------------------
 * UNet input assembly (pipeline:589-593 + scheduler:313-314 + unet:502):
 *   X[(i*T + t)*h*w + p][0:4]  = latents[i % B][t][:, p] / sqrt(sigma^2 + 1)
 *   X[...][4:8]                = (i < B) ? 0 : image_latents[i - B][:, p]        (CFG halves)
 *   X[...][8:Cpad]             = 0   (channel padding so conv_in runs on the MFMA conv kernel)
 *   latents fp32 (B,T,4,h,w); image_latents fp32 (B,4,h,w); X bf16 [2B*T*h*w][Cpad]
 * ---------------------------------------------------------------------------------------------- */
int wiw_prep_unet_input(void* stream, const float* latents, const float* image_latents, int B, int T, int hw,
                        float sigma, int Cpad, void* X);

/* ------------------------------------------------------------------------------------------------
 * CFG combine + Euler step, fp32 (pipeline:606-611, scheduler:635-673, v_prediction):
 *   v  = v_u + g_t (v_c - v_u),  g_t = gmin + (gmax - gmin) * t / (T - 1)
 *   x0 = -sigma/sqrt(sigma^2+1) v + x/(sigma^2+1);   x += (x - x0)/sigma * (sigma_next - sigma)
 *   V fp32 [2B*T*h*w][ldv] (UNet output, channels 0..3); latents fp32 (B,T,4,h,w) updated in place.
 * ---------------------------------------------------------------------------------------------- */
int wiw_cfg_euler_step(void* stream, const float* V, int ldv, float* latents, int B, int T, int hw, float sigma,
                       float sigma_next, float gmin, float gmax);

/* ------------------------------------------------------------------------------------------------
 * Temporal VAE (SURVEY.md §8 rows a6 / a20 — the callers either side of the denoising loop).  The VAE
 * reuses wiw_gemm_bf16 (3x3 / stride-2 / upsample / temporal convolutions, 1x1 shortcuts, attention
 * projections) and the GroupNorm entry points; three operators exist only for it:
 *
 * wiw_softmax_rows_f32_bf16: P[r][0:cols] = softmax(X[r][0:cols]) per row, fp32 in, bf16 out.  Middle of the
 *   single-head, head_dim = C attention of the VAE mid blocks (legacy AttnProcessor of
 *   dp/models/attention_processor.py as instantiated by dp/models/unets/unet_2d_blocks.py UNetMidBlock2D and
 *   unet_3d_blocks.py:930-990 MidBlockTemporalDecoder): scores = scale * Q.K^T and P.V are wiw_gemm_bf16 calls.
 *   cols % 4 == 0, cols <= 16384, ldx / ldp % 4 == 0.
 * wiw_vae_time_conv_out: the decoder's last layer, Conv3d(3, 3, (3,1,1), padding (1,0,0))
 *   (dp/models/autoencoders/autoencoder_kl_temporal_decoder.py:87-161 `time_conv_out`) fused with the
 *   NHWC -> NCHW frame layout:  Y fp32 [frames*HW][ldy] (channels 0..2, ldy % 4 == 0), weight fp32 [3][3][3]
 *   = [co][ci][dt], out fp32 [frames][3][HW]; frames = batch * T, zero padding over T per batch item.
 * wiw_nchw_f32_to_nhwc_bf16: out[(n*HW + p)][c] = c < Cin ? scale * X[n][c][p] : 0  (bf16, Cpad % 8 == 0):
 *   latents / scaling_factor (pipeline:288) or pixels -> the padded NHWC rows conv_in consumes.
 * ---------------------------------------------------------------------------------------------- */
int wiw_softmax_rows_f32_bf16(void* stream, const float* X, int64_t ldx, int64_t rows, int cols, void* P, int64_t ldp);
int wiw_vae_time_conv_out(void* stream, const float* Y, int ldy, const float* weight, const float* bias, int frames,
                          int T, int HW, float* out);
int wiw_nchw_f32_to_nhwc_bf16(void* stream, const float* X, int frames, int Cin, int HW, float scale, int Cpad,
                              void* out);

/* ------------------------------------------------------------------------------------------------
 * CLIP image encoder: `_encode_image` of
 * dp/pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py:183-229.  The transformer's linear layers,
 * LayerNorms and residual adds are wiw_gemm_bf16 (with WIW_EPI_GELU / WIW_EPI_QUICK_GELU for the MLP) and
 * wiw_layernorm_bf16; two operators exist only for it:
 *
 * wiw_clip_preprocess: `_resize_with_antialiasing` (pipeline:643-669: Gaussian blur, reflect padding, then bicubic
 *   interpolation with align_corners=True) to out_size x out_size + `(x + 1) / 2` + (x - mean) / std (pipeline:199-208) +
 *   the im2col of the stride-`patch` patch-embedding convolution (transformers CLIPVisionEmbeddings), written as bf16
 *   operand rows of the patch GEMM:
 *     img  fp32 DEVICE [B][3][H0][W0] in [-1, 1];  tmp fp32 DEVICE scratch [B*3*H0*out_size];
 *     A    bf16 DEVICE [B*rows_per_image][ldA]: pixel (c, y, x) of image b -> row b*rows_per_image + 1 + (y/patch)*(out_size/patch)
 *          + x/patch, column c*patch*patch + (y%patch)*patch + x%patch  (row 0 of an image is the class token's, left untouched,
 *          as are the padding rows / columns: the caller zeroes the buffer once);
 *     taps_x / taps_y (odd counts <= 31), mean[3], inv_std[3]: small HOST arrays (they become kernel arguments).
 * wiw_attn_small_bf16: softmax(Q K^T * scale) V for short sequences and head_dim = any multiple of 16 up to 128
 *   (ViT-H/14: S = 257, 16 heads of 80; F.scaled_dot_product_attention in transformers' CLIPAttention):
 *     QK bf16 [seqs*Sp][ldqk] (Q at column h*head_dim, K at k_col_off + h*head_dim), Vt bf16 [heads*head_dim][ldvt]
 *     (V TRANSPOSED, token index = seq*Sp + s), O bf16 [seqs*Sp][ldo]; Sp = row stride of a sequence, multiple of 16,
 *     >= S; keys >= S are masked, query rows S .. Sp-1 are written as zeros.
 * ---------------------------------------------------------------------------------------------- */
int wiw_clip_preprocess(void* stream, const float* img, int B, int H0, int W0, int out_size, int patch,
                        const float* taps_x, int ntx, const float* taps_y, int nty, const float* mean, const float* inv_std,
                        float* tmp, void* A, int rows_per_image, int ldA);
int wiw_attn_small_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt, void* O,
                        int ldo, int seqs, int S, int Sp, int heads, int head_dim, float scale);

/* Y[c][r] = X[r][c0 + c] for c < C, r < rows (bf16; 64x64 LDS tiles, 128-byte segments on both sides).  Hands
 * wiw_attn_spatial_bf16 its V^T operand from the V columns of ONE fused q|k|v projection (to_q / to_k / to_v of
 * dp/models/attention_processor.py:2358-2366 as a single GEMM).  rows, C, c0, ldx, ldy multiples of 8. */
int wiw_transpose_bf16(void* stream, const void* X, int64_t ldx, int c0, int64_t rows, int C, void* Y, int64_t ldy);

/* ------------------------------------------------------------------------------------------------
 * Fine-tuning step (FTsvd/train_svd.py:844-970) — first entries of that row; the operators' backward kernels are not built.
 * wiw_adamw_step: torch.optim.AdamW (train_svd.py:653, 1123-1130) on a flat fp32 range, ONE pass: decoupled weight decay,
 *   moment updates, bias-corrected step; `p16` (may be NULL) receives the refreshed 16-bit copy the GEMMs read.
 *   All fp32 pointers 16-byte aligned; step counts from 1.
 * wiw_edm_loss_grad: the EDM loss of one sample and dL/d(model_pred) (train_svd.py:940-952):
 *   denoised = v c_out + c_skip noisy, c_out = -s / sqrt(s^2 + 1), c_skip = 1 / (s^2 + 1), w = (1 + s^2) / s^2;
 *   loss = mean(w (denoised - target)^2) = (1 / n) * sum(partial[0 .. n_partial)) (block sums, fixed order);
 *   grad = 2 w c_out (denoised - target) / n.
 * ---------------------------------------------------------------------------------------------- */
int wiw_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step);
int wiw_edm_loss_grad(void* stream, const float* pred, const float* noisy, const float* target, int64_t n, float sigma,
                      float* grad, float* partial, int n_partial);
/* wiw_ema_step_f32: `EMAModel.step` of `--use_ema` (dp/training_utils.py:425-472, train_svd.py:979-980) on a flat fp32 range:
 *   shadow -= one_minus_decay * (shadow - param), the three fp32 operations in that order (bit-identical to the reference's
 *   `s_param.sub_(one_minus_decay * (s_param - param))`).  Pointers 16-byte aligned. */
int wiw_ema_step_f32(void* stream, float* shadow, const float* param, int64_t n, float one_minus_decay);

/* Backward building blocks of the same row (16-bit activations and activation gradients, fp32 parameter gradients; all
 * deterministic: fixed-order partials, no floating-point atomics).  The GEMM-shaped gradients (dX = dY . W, dW = dY^T . X) are
 * wiw_gemm_bf16 launches on transposed operands (wiw_transpose_bf16), see world-in-world_amd/train.py.
 *   wiw_colsum          out[p][c] = sum of the rows [p rpp, min(rows, (p + 1) rpp)), rpp = ceil(rows / parts), of X[rows][C]
 *                       (X 16-bit with C % 8 == 0, or fp32 with is_f32 and C % 4 == 0; 16-byte aligned): bias gradients,
 *                       per-frame sums (parts = frames x k, k | rows per frame, then parts = frames over the partials) and the
 *                       final sum of per-wave partials.
 *   wiw_layernorm_bwd   nn.LayerNorm backward (dp/models/attention.py:659-694 norms): dX (16-bit) and per-wave partials
 *                       partial[w][0][C] = sum dy xhat, partial[w][1][C] = sum dy, w < wiw_layernorm_bwd_partials(rows).
 *   wiw_geglu_bwd       GEGLU backward (activations.py:117-123): P = [v | g] the saved projection output [rows][2 Ch],
 *                       dH [rows][Ch] -> dP [rows][2 Ch]. */
int wiw_colsum(void* stream, const void* X, int is_f32, int64_t rows, int C, int parts, float* out);
int64_t wiw_layernorm_bwd_partials(int64_t rows);
int wiw_layernorm_bwd(void* stream, const void* X, const void* dY, const float* gamma, int64_t rows, int C, float eps,
                      const void* dRes /* or NULL: gradient of the residual path, added to dX */, void* dX, float* partial);
int wiw_geglu_bwd(void* stream, const void* P, const void* dH, int64_t rows, int Ch, void* dP);
/*   wiw_geglu_fwd       H = v * gelu_erf(g) from a SAVED projection output P = [v | g] (the training forward keeps P for
 *                       wiw_geglu_bwd; inference fuses this into the GEMM epilogue on packed weights). */
int wiw_geglu_fwd(void* stream, const void* P, int64_t rows, int Ch, void* H);
/*   wiw_groupnorm_bwd   GroupNorm(32)(+SiLU) backward (dp/models/resnet.py:320-373, 594-631 norms) on [rows][C], units of
 *                       rows_per_unit rows, stats = (mean, variance) per (unit, group) from wiw_groupnorm_stats:
 *                       dX (16-bit); unit_cs[unit][0][C] = sum dz, unit_cs[unit][1][C] = sum dz xhat (dbeta / dgamma = their sums
 *                       over the units: wiw_colsum); scratch: AB [units][32][2], partial [units][splits][2][C] floats with
 *                       splits = ceil(rows_per_unit / rows_per_block). */
/*   wiw_attn_bwd_bf16   self-attention backward (F.scaled_dot_product_attention, attention_processor.py:2383-2385), head_dim 64,
 *                       `seqs` sequences of S rows at row stride Sp (Sp % 16 == 0): QKV [seqs*Sp][ld] with Q | K | V of head h at
 *                       columns h*64, k_off + h*64, v_off + h*64;  Qt, Kt, dOt = transposes [heads*64][ldt] of the Q and K column
 *                       blocks and of dO (wiw_transpose_bf16);  O, dO [rows][ldo];  dQKV [rows][ldd] receives dQ | dK | dV at the
 *                       same column offsets;  lse, dsum: fp32 [seqs*heads*Sp] scratch (row log-sum-exp in the log2 domain and
 *                       D = sum_d dO O, written by the first kernel, read by the second).  Scores are recomputed; deterministic.
 *                       S == Sp, S % 32 == 0, S >= 128 (the spatial sequences): LDS-tiled kernels, 128 rows per workgroup,
 *                       32-deep contraction steps, transposed operands read from the row-major tiles (Qt, Kt, dOt may be
 *                       NULL); otherwise (temporal sequences of 14 padded to 16): one wave per 16-row tile, operands from
 *                       global memory incl. the three transposes. */
int wiw_attn_bwd_bf16(void* stream, const void* QKV, int ld, int k_off, int v_off, const void* Qt, const void* Kt, const void* dOt,
                      int64_t ldt, const void* O, const void* dO, int ldo, void* dQKV, int ldd, float* lse, float* dsum, int seqs,
                      int S, int Sp, int heads, int head_dim, float scale);
/*   wiw_attn_spatial_lse_bf16 / wiw_attn_bwd_given_lse_bf16 (ABI 10): the forward of the fine-tuning step (attention_processor.py:
 *                       2383-2385 under autograd) writes lse[(seq * heads + h) * S + q] = log2 sum_k 2^(s_qk scale log2 e) next to O
 *                       (fp32 sums of the un-rounded probabilities), and the backward takes it instead of recomputing it with a
 *                       Q.K^T pass (spatial sequences: S % 32 == 0, S >= 128; `lse` is read-only there). */
int wiw_attn_spatial_lse_bf16(void* stream, const void* QK, int ldqk, int k_col_off, const void* Vt, int64_t ldvt, void* O, int ldo,
                              int frames, int S, int heads, float scale, const void* zeros, float* lse);
int wiw_attn_bwd_given_lse_bf16(void* stream, const void* QKV, int ld, int k_off, int v_off, const void* O, const void* dO, int ldo,
                                void* dQKV, int ldd, const float* lse, float* dsum, int seqs, int S, int heads, int head_dim,
                                float scale);
/*   wiw_gather_taps_bf16  im2col rows for the weight gradient of the implicit-GEMM convolutions: Xcol[m][tap*C + c] = X[src(m, tap)][c]
 *                       (zeros outside the image / clip); 9 taps (3x3 pad 1 over (H, Wd); stride 2: output (H, Wd), input
 *                       (2H, 2Wd)) or, with temporal != 0, 3 taps over T.
 *   wiw_axpby_bf16      out = a x + b y (y may be NULL): residual joins and gradient fan-in.
 *   wiw_silu_bf16       out = silu(x), or with backward != 0: out = dy silu'(x)  (the embedding MLPs, embeddings.py:804-816).
 *   wiw_dot_bf16        block partials of sum x (y - z) (z may be NULL): the AlphaBlender mix-factor gradient.
 *   wiw_row_map_bf16    structured row re-orderings [rows][C]: 0 nearest upsample x2, 1 dilate x2 (gradient of the stride-2
 *                       conv as a stride-1 conv), 2 sum-pool 2x2 (gradient of the upsample), 3 (b,t,s) -> (b,s,t padded to Tp)
 *                       and 4 back (temporal attention as per-site sequences). */
int wiw_gather_taps_bf16(void* stream, const void* X, int64_t M, int C, int H, int Wd, int T, int temporal, int stride, void* Xcol);
/*   wiw_wgrad_tn_bf16   weight gradient from ROW-MAJOR operands: slabs[s][n][k] = sum over the rows of split s of dY[m][n] X[m][k]
 *                       (fp32 [splits][N][K]; rows split in whole 32-row steps, ceil(M / splits) rounded up to 32; the caller
 *                       sums the slabs in order, e.g. wiw_colsum(slabs, is_f32, rows = splits, C = N*K, parts = 1)).  Both
 *                       MFMA operands are read transposed from LDS (ds_read_b64_tr_b16): no dY^T / X^T copies.  N, K, ldy,
 *                       ldx % 8 == 0; rows >= M and columns >= N / K contribute zeros. */
int wiw_wgrad_tn_bf16(void* stream, const void* dY, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K, int splits,
                      float* slabs);
/*   wiw_gather_taps_t_bf16  the same im2col rows written TRANSPOSED, XcolT [taps*C][Mp] (zero columns for M <= m < Mp, Mp % 8
 *                       == 0): directly the K-contiguous operand of the weight-gradient GEMM, without the [M][taps*C]
 *                       intermediate and its transpose. */
int wiw_gather_taps_t_bf16(void* stream, const void* X, int64_t M, int64_t Mp, int C, int H, int Wd, int T, int temporal, int stride,
                           void* XcolT);
int wiw_axpby_bf16(void* stream, const void* X, const void* Y, float a, float b, int64_t n, void* out);
int wiw_silu_bf16(void* stream, const void* X, const void* dY, int backward, int64_t n, void* out);
int wiw_dot_bf16(void* stream, const void* X, const void* Y, const void* Z, int64_t n, float* partial, int n_partial);
int wiw_row_map_bf16(void* stream, const void* X, int mode, int64_t out_rows, int C, int H, int Wd, int T, int Tp, int S, void* out);
int wiw_groupnorm_bwd(void* stream, const void* X, const void* dY, const float* stats, const float* gamma, const float* beta,
                      int64_t rows, int C, int rows_per_unit, float eps, int silu, void* dX, float* unit_cs, float* AB,
                      float* partial, int rows_per_block);

/* Utility: fill fp32 buffer. */
int wiw_fill_f32(void* stream, float* p, int64_t n, float value);

#ifdef __cplusplus
}
#endif
#endif /* WIW_SVD_H */
