"""ctypes binding of libwiwsvd.so (C ABI: include/wiw_svd.h).

PyTorch supplies device memory (`tensor.data_ptr()`) and the HIP stream; every operator below is a
hand-written gfx950 kernel.  There is NO fallback: a missing library or a failing call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WIW_LIB", os.path.join(_HERE, "libwiwsvd.so"))
LIB_PATH_F16 = os.environ.get("WIW_LIB_F16", os.path.join(_HERE, "libwiwsvd_f16.so"))   # same sources, -DWIW_F16
DTYPE_CODES = {torch.bfloat16: 0, torch.float16: 1}   # wiw_dtype() of the two builds

A_DENSE, A_CONV3X3, A_CONV3X3_S2, A_CONV3X3_UP, A_CONV_T3, A_CONV3X3_S2P = 0, 1, 2, 3, 4, 5
EPI_GEGLU, EPI_SILU, EPI_OUT_F32, EPI_GELU, EPI_QUICK_GELU = 1, 2, 4, 8, 16
W_TILED = 32     # epilogue bit: W is pre-tiled for the LDS-DMA stream (include/wiw_svd.h)
EPI_LNFOLD = 64  # epilogue bit: A is the raw LayerNorm input, W = W * gamma, lnfold = [s | t] (include/wiw_svd.h)
EPI_RES1_F32, EPI_RES2_F32 = 128, 256   # epilogue bits: res1 / res2 are fp32 (the fp32 residual stream, ABI 11)
K_HALO32 = 1024  # conv3x3: halo-staged A operand, k = ((c / 32) * 9 + tap) * 32 + c % 32 (include/wiw_svd.h, conv_halo_ok below)
K_CMAJOR = 512   # conv modes: K of W is channel-block major, k = ((c / 64) * taps + tap) * 64 + c % 64 (include/wiw_svd.h)
GEGLU_TILE = 80  # value|gate half-tile width of the packed GEGLU weights (gemm.hip BN / 2)
FFN_CHUNK = 64   # hidden units per chunk of the fused FeedForward kernel (ffn.hip): W1 rows in chunks of [64 value | 64 gate]
FFN_C, FFN_HIDDEN = 320, 1280   # the one shape wiw_ffn_geglu_bf16 is built for
_GN_RPB_R5 = bool(os.environ.get("WIW_GN_RPB_R5"))
PROF_RES1, PROF_RES2 = 1 << 24, 1 << 25   # bench.py profile keys only: the launch read a res1 / res2 operand


class TiledW:
    """A static GEMM weight [N][K] (bf16) re-laid-out ONCE for the kernels' LDS-DMA stream: ceil(N/8) x (K/64) blocks of
    1 KiB, block (nb, kt) = rows 8nb..8nb+7 x k-values 64kt..64kt+63, row r of a block at bytes r*128.. with its eight
    16-byte chunks XOR-swizzled (position p holds chunk p ^ r) — exactly the LDS image of gemm.hip, so one DMA
    instruction copies one contiguous KiB: 63 B/clk/CU instead of 25 for eight row segments K*2 bytes apart
    (tools/ubench/lds_fill.hip).  Quacks like the [N, K] tensor it replaces (`shape`, `data_ptr`)."""

    def __init__(self, w: torch.Tensor, sw16: bool = False):
        """sw16: the chunk swizzle of the 32x32x16-MFMA kernels (csrc/ffn32.hip): position p of row R holds chunk
        p ^ ((R >> 1) & 7) — 16 rows distinct mod 16 then cover all LDS banks once per ds_read_b128 (the default swizzle,
        p ^ (R & 7), serves the 16x16x32 fragment reads of gemm.hip)."""
        assert w.dim() == 2 and w.dtype in DTYPE_CODES and w.shape[1] % 64 == 0, "TiledW: 16-bit [N, K] with K % 64 == 0"
        self.shape = tuple(w.shape)
        self.sw16 = bool(sw16)
        self.data = tile_weight(w, sw16=sw16)

    def data_ptr(self):
        return self.data.data_ptr()

    def untiled(self) -> torch.Tensor:
        return untile_weight(self.data, *self.shape, sw16=self.sw16)


def _tile_swizzle(Np: int, device, sw16: bool) -> torch.Tensor:
    """[nb][r][p] -> the chunk stored at position p of row 8 nb + r."""
    R = torch.arange(Np, device=device).reshape(Np // 8, 8)
    sw = ((R >> 1) & 7) if sw16 else (R & 7)
    return torch.arange(8, device=device)[None, None, :] ^ sw[:, :, None]


def tile_weight(w: torch.Tensor, sw16: bool = False) -> torch.Tensor:
    """[N, K] bf16 -> flat tiled tensor (layout: class TiledW)."""
    N, K = w.shape
    Np = -(-N // 8) * 8
    if Np != N:
        w = torch.cat([w, w.new_zeros(Np - N, K)])
    x = w.reshape(Np // 8, 8, K // 64, 8, 8).permute(0, 2, 1, 3, 4)           # [nb][kt][r][chunk][e]
    idx = _tile_swizzle(Np, w.device, sw16)                                   # [nb][r][p]
    x = torch.gather(x, 3, idx[:, None, :, :, None].expand(x.shape[0], x.shape[1], 8, 8, 8))
    return x.contiguous().reshape(-1)


def untile_weight(t: torch.Tensor, N: int, K: int, sw16: bool = False) -> torch.Tensor:
    Np = -(-N // 8) * 8
    x = t.reshape(Np // 8, K // 64, 8, 8, 8)
    idx = _tile_swizzle(Np, t.device, sw16)                                   # the swizzle is an involution per row
    x = torch.gather(x, 3, idx[:, None, :, :, None].expand(x.shape[0], x.shape[1], 8, 8, 8))
    return x.permute(0, 2, 1, 3, 4).reshape(Np, K)[:N].contiguous()


class WiwGemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("out", C.c_void_p),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("res1", C.c_void_p), ("res2", C.c_void_p),
        ("zeros", C.c_void_p), ("A3", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("C1", C.c_int32), ("C2", C.c_int32), ("C3", C.c_int32), ("mode", C.c_int32),
        ("H", C.c_int32), ("Wd", C.c_int32), ("T", C.c_int32),
        ("ldo", C.c_int32), ("ldr1", C.c_int32), ("ldr2", C.c_int32), ("n_out", C.c_int32),
        ("rowvec_ld", C.c_int32), ("rows_per_vec", C.c_int32),
        ("alpha", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
        ("epilogue", C.c_int32), ("splitk", C.c_int32), ("workspace", C.c_void_p),
        ("lnfold", C.c_void_p), ("ln_eps", C.c_float),
    ]


EXPORTS = {
    "wiw_abi_version": (C.c_int, []),
    "wiw_dtype": (C.c_int, []),
    "wiw_last_error": (C.c_char_p, []),
    "wiw_device_check": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "wiw_gemm_bf16": (C.c_int, [C.c_void_p, C.POINTER(WiwGemmArgs)]),
    "wiw_attn_spatial_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "wiw_attn_spatial_ps_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "wiw_attn_spatial_lse_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "wiw_attn_bwd_given_lse_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_float]),
    "wiw_attn_temporal_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_float]),
    "wiw_temporal_attn_block_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "wiw_ffn_geglu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                     C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float]),
    "wiw_ffn32_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                  C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "wiw_ffn_geglu_f32stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                     C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "wiw_ffn_geglu_f32stream2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                      C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int]),
    "wiw_cross_attn_fewkeys_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64,
                                              C.c_int, C.c_int, C.c_int, C.c_float]),
    "wiw_clip_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "wiw_attn_small_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "wiw_groupnorm_scratch_floats": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "wiw_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "wiw_groupnorm_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_void_p]),
    "wiw_groupnorm_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    "wiw_groupnorm_apply_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "wiw_layernorm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wiw_groupnorm_stats_f32in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "wiw_groupnorm_counters": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "wiw_conv_halo_ok": (C.c_int, [C.c_void_p]),
    "wiw_groupnorm_onepass_ok": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_int]),
    "wiw_groupnorm_onepass": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_float, C.c_int, C.c_void_p]),
    "wiw_groupnorm_apply_stats_f32in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "wiw_layernorm_f32in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "wiw_cast_f32_to_16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "wiw_calib_mfma": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "wiw_calib_mfma_random": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "wiw_emb_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p]),
    "wiw_prep_unet_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_int, C.c_void_p]),
    "wiw_cfg_euler_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_float, C.c_float, C.c_float]),
    "wiw_transpose_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int64]),
    "wiw_ema_step_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]),
    "wiw_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]),
    "wiw_edm_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_int]),
    "wiw_gather_taps_t_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]),
    "wiw_wgrad_tn_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]),
    "wiw_colsum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "wiw_layernorm_bwd_partials": (C.c_int64, [C.c_int64]),
    "wiw_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "wiw_geglu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "wiw_groupnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "wiw_gather_taps_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p]),
    "wiw_axpby_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_void_p]),
    "wiw_silu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "wiw_dot_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]),
    "wiw_row_map_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "wiw_geglu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "wiw_attn_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_float]),
    "wiw_fill_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]),
    "wiw_softmax_rows_f32_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64]),
    "wiw_vae_time_conv_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "wiw_nchw_f32_to_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                            C.c_void_p]),
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the C-ABI library and declare every prototype of include/wiw_svd.h.  Raises if absent."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is required (build it with "
            f"`python world-in-world_amd/build.py`); there is no CPU or PyTorch fallback")
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Hip:
    """Thin operator layer over the C ABI: argument checking + pointer marshalling only."""

    def __init__(self, device: torch.device, dtype: torch.dtype = torch.bfloat16):
        """dtype: the 16-bit storage type of activations and weights — torch.bfloat16 (libwiwsvd.so) or torch.float16
        (libwiwsvd_f16.so, the reference's served default).  One Hip serves one dtype; every tensor handed to it must
        have that dtype (16-bit operands) or fp32 (vectors, latents)."""
        if dtype not in DTYPE_CODES:
            raise ValueError("Hip: dtype must be torch.bfloat16 or torch.float16")
        self.dtype = dtype
        self.lib = load_library(LIB_PATH if dtype == torch.bfloat16 else LIB_PATH_F16)
        if self.lib.wiw_abi_version() != 17:
            raise RuntimeError("libwiwsvd ABI version mismatch")
        if self.lib.wiw_dtype() != DTYPE_CODES[dtype]:
            raise RuntimeError("the loaded library was built for the other 16-bit type (wiw_dtype mismatch)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP path needs a ROCm device (torch device type 'cuda')")
        name = C.create_string_buffer(64)
        rc = self.lib.wiw_device_check(self.device.index or 0, name, 64)
        if rc != 0:
            raise RuntimeError(f"wiw_device_check: {self.lib.wiw_last_error().decode()}")
        self.arch = name.value.decode()
        # HIP's current device is PER THREAD: grids are sized from the current device's CU count, streams belong to
        # self.device.  One process serves one GPU (make it current here); server threads call bind_thread().
        torch.cuda.set_device(self.device)
        self.zeros = torch.zeros(16384 + 64, dtype=torch.uint8, device=self.device)   # the kernels' zero page (include/wiw_svd.h)
        # bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream:
        # entries are (start_event, end_event, algorithmic_flops, mode, (M, N, K, epilogue))
        self.gemm_profile = None
        self._splitk_ws = None
        self.gn_two_kernels = bool(os.environ.get("WIW_GN_TWO_KERNELS"))     # A/B knob: statistics + apply launches at every level
        # int32 counters of the GroupNorm statistics launches, one row per STREAM that ever launched one (see _gn_buffers);
        # all rows are allocated and zeroed here: handing one out is legal inside a hipGraph capture
        # 40 rows: torch hands out at most 32 pooled streams + the default stream + the graph-capture stream (10 MB)
        self._gn_cnt_pool = torch.zeros(40, 65536, dtype=torch.int32, device=self.device)
        self._gn_cnt = {}
        self._warmup_stream = None
        # ... and this one for the non-GEMM kernels: (start_event, end_event, family, algorithmic_flops, algorithmic_bytes)
        self.kernel_profile = None

    # ---- helpers
    def bind_thread(self) -> None:
        """Make this library's device the calling thread's current HIP device (handler threads of serve_tcp start on
        device 0 whatever --device says)."""
        torch.cuda.set_device(self.device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def warmup_stream(self):
        """The ONE side stream graph captures warm up on (pipeline.GraphedForward): stream-keyed state (the GroupNorm counter
        rows) then sees three streams per process — eager, warm-up, capture — however many shapes are captured."""
        if self._warmup_stream is None:
            self._warmup_stream = torch.cuda.Stream(self.device)
        return self._warmup_stream

    def _timed(self, family: str, flops: float, nbytes: float, fn):
        """Run `fn` (one or more launches of one kernel family) between two HIP events when bench.py profiles."""
        if self.kernel_profile is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.kernel_profile.append((e0, e1, family, float(flops), float(nbytes)))
        return r

    def _ck(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.wiw_last_error().decode()}")

    # ---- operators
    @staticmethod
    def conv_halo_ok(M: int, N: int, C1: int, H: int, Wd: int, up: bool = False) -> bool:
        """Geometry the halo-staged 3x3 convolution kernel takes (WIW_K_HALO32; the C side re-checks: wiw_conv_halo_ok):
        a 256-row tile is whole image rows, of at most two frames (up = behind the nearest x2 upsample, H x Wd the output
        size: of one frame, Wd >= 64)."""
        if M % 256 or M % (H * Wd) or C1 % 64 or N % 320:
            return False
        if up:
            return Wd in (64, 128) and H % 2 == 0 and (H * Wd) % 256 == 0
        return Wd in (32, 64, 128) and H >= 256 // Wd

    def gemm(self, A, W, out, *, M, N, K, C1, A2=None, C2=0, A3=None, C3=0, mode=A_DENSE, H=0, Wd=0, T=0, bias=None, rowvec=None,
             rowvec_ld=0, rows_per_vec=1, res1=None, ldr1=0, beta1=0.0, res2=None, ldr2=0, beta2=0.0, alpha=1.0,
             ldo=None, epilogue=0, n_out=0, splitk=1, lnfold=None, ln_eps=1e-5):
        """lnfold: fp32 [2, N] = (s | t) — A is then the RAW input of a LayerNorm(eps = ln_eps) whose affine is folded into
        W (= W * gamma) and t (EPI_LNFOLD, include/wiw_svd.h); no LayerNorm pass runs."""
        a = WiwGemmArgs()
        a.A, a.A2, a.W, a.out = _p(A), _p(A2), _p(W), _p(out)
        a.bias, a.rowvec, a.res1, a.res2 = _p(bias), _p(rowvec), _p(res1), _p(res2)
        a.zeros = self.zeros.data_ptr()
        a.A3 = _p(A3)
        a.M, a.N, a.K, a.C1, a.C2, a.C3, a.mode = M, N, K, C1, C2, C3, mode
        a.H, a.Wd, a.T = H, Wd, T
        a.ldo = ldo if ldo is not None else (n_out if epilogue & EPI_GEGLU else N)
        a.ldr1, a.ldr2, a.n_out = ldr1, ldr2, n_out
        a.rowvec_ld, a.rows_per_vec = rowvec_ld, rows_per_vec
        a.alpha, a.beta1, a.beta2 = alpha, beta1, beta2
        a.epilogue = epilogue | (W_TILED if isinstance(W, TiledW) else 0)
        # fp32 residual stream (ABI 11): an fp32 res1 / res2 tensor sets its bit; an fp32 `out` needs EPI_OUT_F32 from the caller
        if res1 is not None and res1.dtype == torch.float32:
            a.epilogue |= EPI_RES1_F32
        if res2 is not None and res2.dtype == torch.float32:
            a.epilogue |= EPI_RES2_F32
        assert (out.dtype == torch.float32) == bool(a.epilogue & EPI_OUT_F32), "gemm: fp32 out <=> EPI_OUT_F32"
        if lnfold is not None:
            a.epilogue |= EPI_LNFOLD
            a.lnfold, a.ln_eps = lnfold.data_ptr(), ln_eps
        if splitk > 1:   # fp32 partial sums of the K ranges: one grow-only scratch buffer per Hip (stream-ordered reuse)
            need = splitk * M * N
            if self._splitk_ws is None or self._splitk_ws.numel() < need:
                self._splitk_ws = torch.empty(need, dtype=torch.float32, device=self.device)
            a.splitk, a.workspace = splitk, self._splitk_ws.data_ptr()
        if self.gemm_profile is None:
            self._ck(self.lib.wiw_gemm_bf16(self._stream(), C.byref(a)), "wiw_gemm_bf16")
            return out
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._ck(self.lib.wiw_gemm_bf16(self._stream(), C.byref(a)), "wiw_gemm_bf16")
        e1.record()
        # (profile key: the caller's epilogue bits + which residual operands the launch read, PROF_RES1 / PROF_RES2)
        self.gemm_profile.append((e0, e1, 2.0 * M * N * K, mode,
                                  (M, N, K, epilogue | (PROF_RES1 if res1 is not None else 0) | (PROF_RES2 if res2 is not None else 0)
                                   | (a.epilogue & (EPI_RES1_F32 | EPI_RES2_F32)))))
        return out

    def attn_spatial(self, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads, scale, lse=None):
        """lse (training forward): fp32 [frames * heads * S], receives the row log-sum-exp (log2 domain) for `attn_backward`."""
        # algorithmic work: Q.K^T and P.V, 2*S*S*64 each per (frame, head); bytes: Q, K, V read + O written (bf16)
        if lse is not None:
            assert lse.dtype == torch.float32 and lse.numel() == frames * heads * S
            self._timed("attn_spatial", 4.0 * frames * heads * S * S * 64, 8.0 * frames * S * heads * 64, lambda: self._ck(
                self.lib.wiw_attn_spatial_lse_bf16(self._stream(), _p(QK), ldqk, k_col_off, _p(Vt), ldvt, _p(O), ldo, frames, S,
                                                   heads, scale, self.zeros.data_ptr(), _p(lse)), "wiw_attn_spatial_lse_bf16"))
            return O
        self._timed("attn_spatial", 4.0 * frames * heads * S * S * 64, 8.0 * frames * S * heads * 64, lambda: self._ck(
            self.lib.wiw_attn_spatial_bf16(self._stream(), _p(QK), ldqk, k_col_off, _p(Vt), ldvt, _p(O), ldo,
                                           frames, S, heads, scale, self.zeros.data_ptr()), "wiw_attn_spatial_bf16"))
        return O

    def attn_spatial_ps(self, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, frames, S, heads):
        """Spatial attention on a Q that the projection pre-scaled by ATTN_PRESCALE = log2(e) / sqrt(64) (csrc/attention32.hip)."""
        self._timed("attn_spatial", 4.0 * frames * heads * S * S * 64, 8.0 * frames * S * heads * 64, lambda: self._ck(
            self.lib.wiw_attn_spatial_ps_bf16(self._stream(), _p(QK), ldqk, k_col_off, _p(Vt), ldvt, _p(O), ldo,
                                              frames, S, heads, self.zeros.data_ptr()), "wiw_attn_spatial_ps_bf16"))
        return O

    def attn_temporal(self, QKV, ldqkv, O, ldo, batch, T, S, heads, scale):
        self._timed("attn_temporal", 4.0 * batch * S * heads * T * T * 64, 8.0 * batch * T * S * heads * 64, lambda: self._ck(
            self.lib.wiw_attn_temporal_bf16(self._stream(), _p(QKV), ldqkv, _p(O), ldo, batch, T, S, heads, scale),
            "wiw_attn_temporal_bf16"))
        return O

    def temporal_attn_block(self, X, Wqkv, fold, O, ldo, batch, T, S, heads, eps, scale):
        """LayerNorm (folded) + per-head QKV projection + temporal attention in one kernel (temporal.hip)."""
        Cn, M = heads * 64, batch * T * S
        # algorithmic work: the Q/K/V projection (2*M*3C*C) + the attention core at the real T; bytes: X read once,
        # O written, weights once (bf16)
        self._timed("temporal_block", 2.0 * M * 3 * Cn * Cn + 4.0 * batch * S * heads * T * T * 64,
                    2.0 * (2 * M * Cn + 3 * Cn * Cn), lambda: self._ck(
            self.lib.wiw_temporal_attn_block_bf16(self._stream(), _p(X), _p(Wqkv), _p(fold), _p(O), ldo, batch, T, S,
                                                  heads, eps, scale, self.zeros.data_ptr()),
            "wiw_temporal_attn_block_bf16"))
        return O

    def ffn_geglu(self, X, W1, b1, W2, b2, out, M, *, ldx=FFN_C, rowvec=None, rowvec_ld=0, rows_per_vec=1, res1=None, ldr1=0,
                  beta1=0.0, res2=None, ldr2=0, beta2=0.0, alpha=1.0, ldo=FFN_C, ln=False, ln_eps=1e-5, out16=None):
        """Fused (LayerNorm +) GEGLU FeedForward of the 320-channel level (ffn.hip): the [M, 1280] hidden tensor never
        exists.  W1 / b1 packed by `unet.pack_geglu(..., tile=FFN_CHUNK)` (+ TiledW), W2 a TiledW of [320, 1280].
        ABI 16: an fp32 X (the residual stream itself; needs ln=True) and `out16`, a second 16-bit copy of the output."""
        flops = 2.0 * M * (2 * FFN_HIDDEN * FFN_C + FFN_C * FFN_HIDDEN)
        el = lambda t: 0 if t is None else t.element_size()   # noqa: E731
        nbytes = 1.0 * M * FFN_C * (el(X) + el(out) + el(res1) + el(res2) + el(out16)) + 2.0 * 3 * FFN_HIDDEN * FFN_C

        f32 = ((out.dtype == torch.float32) * 1 + (res1 is not None and res1.dtype == torch.float32) * 2
               + (res2 is not None and res2.dtype == torch.float32) * 4 + (X.dtype == torch.float32) * 8)
        assert not (f32 & 8) or ln, "ffn_geglu: an fp32 X is read by the fused LayerNorm only"
        assert out16 is None or out16.dtype == self.dtype

        def launch():
            if (f32 & 8) or out16 is not None:
                self._ck(self.lib.wiw_ffn_geglu_f32stream2(self._stream(), _p(X), ldx, _p(W1), _p(b1), _p(W2), _p(b2), _p(rowvec),
                                                           rowvec_ld, rows_per_vec, _p(res1), ldr1, beta1, _p(res2), ldr2, beta2,
                                                           alpha, _p(out), ldo, M, FFN_C, FFN_HIDDEN, 1 if ln else 0, ln_eps, f32,
                                                           _p(out16), FFN_C if out16 is not None else 0),
                         "wiw_ffn_geglu_f32stream2")
                return
            if f32:      # fp32 residual stream: the F32E instantiation (fragment-layout epilogue, one rounding)
                self._ck(self.lib.wiw_ffn_geglu_f32stream(self._stream(), _p(X), ldx, _p(W1), _p(b1), _p(W2), _p(b2), _p(rowvec),
                                                          rowvec_ld, rows_per_vec, _p(res1), ldr1, beta1, _p(res2), ldr2, beta2,
                                                          alpha, _p(out), ldo, M, FFN_C, FFN_HIDDEN, 1 if ln else 0, ln_eps, f32),
                         "wiw_ffn_geglu_f32stream")
                return
            self._ck(self.lib.wiw_ffn_geglu_bf16(self._stream(), _p(X), ldx, _p(W1), _p(b1), _p(W2), _p(b2), _p(rowvec),
                                                 rowvec_ld, rows_per_vec, _p(res1), ldr1, beta1, _p(res2), ldr2, beta2, alpha,
                                                 _p(out), ldo, M, FFN_C, FFN_HIDDEN, 1 if ln else 0, ln_eps),
                     "wiw_ffn_geglu_bf16")
        self._timed("ffn_fused", flops, nbytes, launch)
        return out

    def ffn32_geglu(self, X, W1, b1, W2, b2, out, M, *, ldx=FFN_C, rowvec=None, rowvec_ld=0, rows_per_vec=1, res1=None, ldr1=0,
                    beta1=0.0, res2=None, ldr2=0, beta2=0.0, alpha=1.0, ldo=FFN_C, ln=False, ln_eps=1e-5):
        """The same operator on the 32x32x16-MFMA kernel (csrc/ffn32.hip, round 5).  Operands from `unet.pack_ffn32`: W1 / W2 are
        TiledW(sw16=True), W1 / b1 packed in chunks of [32 value | 32 gate] rows, the value half of b1 pre-multiplied by 0.5."""
        flops = 2.0 * M * (2 * FFN_HIDDEN * FFN_C + FFN_C * FFN_HIDDEN)
        nbytes = 2.0 * M * FFN_C * (2 + (res1 is not None) + (res2 is not None)) + 2.0 * 3 * FFN_HIDDEN * FFN_C
        f32 = ((out.dtype == torch.float32) * 1 + (res1 is not None and res1.dtype == torch.float32) * 2
               + (res2 is not None and res2.dtype == torch.float32) * 4)
        assert getattr(W1, "sw16", False) and getattr(W2, "sw16", False), "ffn32_geglu: weights must be TiledW(sw16=True) (unet.pack_ffn32)"
        self._timed("ffn_fused", flops, nbytes, lambda: self._ck(
            self.lib.wiw_ffn32_geglu(self._stream(), _p(X), ldx, _p(W1), _p(b1), _p(W2), _p(b2), _p(rowvec), rowvec_ld, rows_per_vec,
                                     _p(res1), ldr1, beta1, _p(res2), ldr2, beta2, alpha, _p(out), ldo, M, FFN_C, FFN_HIDDEN,
                                     1 if ln else 0, ln_eps, f32), "wiw_ffn32_geglu"))
        return out

    def cross_attn_fewkeys(self, Q, ldq, K, V, O, ldo, rows, rows_per_item, heads, P, scale):
        """Cross-attention over P <= 8 conditioning tokens per item (csrc/cross_attn.hip): Q / O [rows, heads*64] 16-bit,
        K / V [rows / rows_per_item, P, heads*64] 16-bit contiguous."""
        C_ = heads * 64
        assert K.is_contiguous() and V.is_contiguous() and K.shape == V.shape == (rows // rows_per_item, P, C_)
        self._timed("attn_cross", 4.0 * rows * P * C_, 4.0 * rows * C_, lambda: self._ck(
            self.lib.wiw_cross_attn_fewkeys_bf16(self._stream(), _p(Q), ldq, _p(K), _p(V), _p(O), ldo, rows, rows_per_item, heads, P,
                                                 scale), "wiw_cross_attn_fewkeys_bf16"))
        return O

    def attn_small(self, QK, ldqk, k_col_off, Vt, ldvt, O, ldo, seqs, S, Sp, heads, head_dim, scale):
        self._timed("attn_small", 4.0 * seqs * heads * S * S * head_dim, 8.0 * seqs * S * heads * head_dim, lambda: self._ck(
            self.lib.wiw_attn_small_bf16(self._stream(), _p(QK), ldqk, k_col_off, _p(Vt), ldvt, _p(O), ldo, seqs, S, Sp,
                                         heads, head_dim, scale), "wiw_attn_small_bf16"))
        return O

    def clip_preprocess(self, img, B, H0, W0, out_size, patch, taps_x, taps_y, mean, inv_std, tmp, A, rows_per_image, ldA):
        fa = lambda v: (C.c_float * len(v))(*[float(x) for x in v])   # noqa: E731  small HOST parameter arrays
        self._ck(self.lib.wiw_clip_preprocess(self._stream(), _p(img), B, H0, W0, out_size, patch, fa(taps_x), len(taps_x),
                                              fa(taps_y), len(taps_y), fa(mean), fa(inv_std), _p(tmp), _p(A), rows_per_image,
                                              ldA), "wiw_clip_preprocess")
        return A

    @staticmethod
    def gn_rows_per_block(rows_per_unit: int, clip: bool) -> int:
        """Block size of the statistics pass — a function of the unit size and the KIND of norm only (never of the
        batch): per-frame norms (28+ units per request) use large blocks, per-clip norms (one unit per CFG item:
        TemporalResnetBlock) split a unit into ~500 blocks."""
        # round 6 sweep on the final library (profiles/r20r_gn_stats_rpb_sweep.txt, 16-bit input, statistics pass alone): larger
        # blocks win at the two outer levels — per-frame 72x128: 128 -> 256 rows 38.7 -> 32.7 us (C = 320), 65.6 -> 59.1 (C = 640);
        # per-frame 36x64: 64 -> 128 rows 23.3 -> 21.0 us (C = 640), 34.7 -> 32.5 (C = 1280; 256 is slower there); per-clip 36x64
        # (32 256 rows): 64 -> 128 rows 28.0 -> 25.9 us.  The inner levels run the one-pass kernel and do not come here.
        if _GN_RPB_R5:      # A/B knob WIW_GN_RPB_R5=1: the block sizes of rounds 4-5
            if not clip:
                return 128 if rows_per_unit >= 8192 else (64 if rows_per_unit >= 2048 else (32 if rows_per_unit >= 512 else 16))
            rpb = 16
            while rpb < 256 and rpb * 2 * 448 <= rows_per_unit:
                rpb *= 2
            return rpb
        if not clip:
            return 256 if rows_per_unit >= 8192 else (128 if rows_per_unit >= 2048 else (32 if rows_per_unit >= 512 else 16))
        rpb = 16
        while rpb < 256 and rpb * 448 <= rows_per_unit:
            rpb *= 2
        return rpb

    def _gn_buffers(self, rows, rows_per_unit, rpb):
        """(stats [units*64], scratch, counters) of wiw_groupnorm_stats.  stats / scratch are fully written by the call
        (deterministic two-level reduction inside the statistics launch).  counters: zero before the first call, left at
        zero by every call — ONE buffer per Hip, made at construction (never inside a graph capture): the GroupNorm
        launches of a Hip are stream-ordered (one request at a time per GPU; the capture's warm-up stream is joined
        before the capture starts)."""
        units = rows // rows_per_unit
        n = int(self.lib.wiw_groupnorm_scratch_floats(rows, rows_per_unit, rpb))
        # The last-block-done reduction (norm.hip) needs its counters at zero before a launch and leaves them at zero: two
        # statistics launches must never overlap on ONE counter buffer.  Launches of one stream are ordered, so the buffer is
        # keyed by stream (the VAE shares this Hip with the UNet; a capture's warm-up runs on a side stream) — ADVICE r4.
        cnt = self._gn_counter_row()
        assert int(self.lib.wiw_groupnorm_counters(rows, rows_per_unit, rpb)) <= cnt.numel(), "groupnorm: more parts than counters"
        buf = torch.empty(units * 64 + n, dtype=torch.float32, device=self.device)
        return buf[: units * 64], buf[units * 64:], cnt

    def _gn_counter_row(self):
        cnt = self._gn_cnt.get(self._stream())
        if cnt is None:
            if len(self._gn_cnt) >= self._gn_cnt_pool.shape[0]:
                raise RuntimeError("groupnorm: more streams than counter rows (Hip._gn_cnt_pool)")
            cnt = self._gn_cnt[self._stream()] = self._gn_cnt_pool[len(self._gn_cnt)]
        return cnt

    def gn_counters_clean(self) -> bool:
        """Debug check (tests): every statistics launch left its counters at zero."""
        torch.cuda.synchronize(self.device)
        return int(self._gn_cnt_pool.abs().max()) == 0

    def groupnorm(self, X1, C1, X2, C2, rows, rows_per_unit, gamma, beta, eps, silu, out=None, clip=False, raw16=None):
        """statistics (deterministic, no atomics) -> fused finalize + apply; returns the normalised (and SiLU'd) bf16
        tensor [rows, C1+C2].  clip=True: the unit is a whole clip (T frames), see gn_rows_per_block.
        fp32 residual stream (ABI 11): fp32 X1 (and X2) select the fp32-input kernels; `raw16` (16-bit [rows, C1 + C2]) then
        also receives the rounded raw input (the operand of the 1x1 conv_shortcut)."""
        f32in = X1.dtype == torch.float32
        assert X2 is None or (X2.dtype == torch.float32) == f32in, "groupnorm: X1 and X2 must have one dtype"
        assert raw16 is None or f32in, "groupnorm: raw16 only with fp32 inputs"
        if os.environ.get("WIW_GN_UNFUSED") and not f32in:   # A/B knob for profiling
            return self.groupnorm_unfused(X1, C1, X2, C2, rows, rows_per_unit, gamma, beta, eps, silu, out, clip)
        Ct = C1 + C2
        if (not f32in and not clip and not self.gn_two_kernels
                and self.lib.wiw_groupnorm_onepass_ok(C1, C2, rows, rows_per_unit)):      # ABI 15: the two inner levels in ONE pass
            if out is None:
                out = torch.empty((rows, Ct), dtype=self.dtype, device=self.device)
            s = self._stream()
            self._timed("groupnorm", 0.0, 4.0 * rows * Ct, lambda: self._ck(self.lib.wiw_groupnorm_onepass(
                s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, _p(gamma), _p(beta), eps, 1 if silu else 0, out.data_ptr()),
                "wiw_groupnorm_onepass"))
            return out
        rpb = self.gn_rows_per_block(rows_per_unit, clip)
        stats, scratch, cnt = self._gn_buffers(rows, rows_per_unit, rpb)
        if out is None:
            out = torch.empty((rows, Ct), dtype=self.dtype, device=self.device)
        s = self._stream()

        def launch():
            fn = self.lib.wiw_groupnorm_stats_f32in if f32in else self.lib.wiw_groupnorm_stats
            self._ck(fn(s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, rpb, stats.data_ptr(), scratch.data_ptr(), cnt.data_ptr()),
                     "wiw_groupnorm_stats")
            if f32in:
                self._ck(self.lib.wiw_groupnorm_apply_stats_f32in(s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, stats.data_ptr(),
                                                                  _p(gamma), _p(beta), eps, 1 if silu else 0, out.data_ptr(),
                                                                  _p(raw16)), "wiw_groupnorm_apply_stats_f32in")
                return
            self._ck(self.lib.wiw_groupnorm_apply_stats(s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, stats.data_ptr(),
                                                        _p(gamma), _p(beta), eps, 1 if silu else 0, out.data_ptr()),
                     "wiw_groupnorm_apply_stats")

        # algorithmic bytes: the input is read twice (statistics, apply) and the output written once (bf16)
        self._timed("groupnorm", 0.0, ((10.0 if f32in else 6.0) + (2.0 if raw16 is not None else 0.0)) * rows * Ct, launch)
        return out

    def groupnorm_unfused(self, X1, C1, X2, C2, rows, rows_per_unit, gamma, beta, eps, silu, out=None, clip=False):
        """stats -> finalize -> apply through the three separate entry points (kept for the ABI tests)."""
        Ct = C1 + C2
        units = rows // rows_per_unit
        rpb = self.gn_rows_per_block(rows_per_unit, clip)
        stats, scratch, cnt = self._gn_buffers(rows, rows_per_unit, rpb)
        ab = torch.empty(units * 2 * Ct, dtype=torch.float32, device=self.device)
        if out is None:
            out = torch.empty((rows, Ct), dtype=self.dtype, device=self.device)
        s = self._stream()
        self._ck(self.lib.wiw_groupnorm_stats(s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, rpb, stats.data_ptr(),
                                              scratch.data_ptr(), cnt.data_ptr()), "wiw_groupnorm_stats")
        self._ck(self.lib.wiw_groupnorm_finalize(s, stats.data_ptr(), _p(gamma), _p(beta), units, Ct, rows_per_unit,
                                                 eps, ab.data_ptr()), "wiw_groupnorm_finalize")
        self._ck(self.lib.wiw_groupnorm_apply(s, _p(X1), C1, _p(X2), C2, rows, rows_per_unit, ab.data_ptr(),
                                              1 if silu else 0, out.data_ptr()), "wiw_groupnorm_apply")
        return out

    def layernorm(self, X, rows, Cn, gamma, beta, eps=1e-5, addvec=None, addvec_ld=0, rows_per_vec=1, sum_out=None,
                  out=None):
        if out is None:
            out = torch.empty((rows, Cn), dtype=self.dtype, device=self.device)
        if X.dtype == torch.float32:    # fp32 residual stream (ABI 11)
            assert addvec is None and sum_out is None, "layernorm: the fp32-input kernel takes no pre-add"
            self._timed("layernorm", 0.0, 6.0 * rows * Cn, lambda: self._ck(
                self.lib.wiw_layernorm_f32in(self._stream(), _p(X), rows, Cn, _p(gamma), _p(beta), eps, out.data_ptr()),
                "wiw_layernorm_f32in"))
            return out
        self._timed("layernorm", 0.0, (6.0 if sum_out is not None else 4.0) * rows * Cn, lambda: self._ck(
            self.lib.wiw_layernorm_bf16(self._stream(), _p(X), rows, Cn, _p(gamma), _p(beta), eps, _p(addvec),
                                        addvec_ld, rows_per_vec, _p(sum_out), out.data_ptr()), "wiw_layernorm_bf16"))
        return out

    def cast16(self, X, out=None):
        """fp32 tensor -> this library's 16-bit type (the MFMA operand of an fp32 residual-stream tensor)."""
        assert X.dtype == torch.float32 and X.numel() % 8 == 0
        if out is None:
            out = torch.empty(X.shape, dtype=self.dtype, device=self.device)
        self._timed("cast16", 0.0, 6.0 * X.numel(), lambda: self._ck(
            self.lib.wiw_cast_f32_to_16(self._stream(), _p(X), X.numel(), out.data_ptr()), "wiw_cast_f32_to_16"))
        return out

    def calibrate_box(self, target_ms: float = 50.0, sustain_s: float = 1.0):
        """What THIS box gives (bench.py `box`): a pure-MFMA launch of ~target_ms (register operands, one 8-wave block per CU),
        the same loop on random operands held for `sustain_s` seconds, and a 1 GiB device copy, all timed with events.
        -> dict(mfma_tflops, mfma_sustained_tflops, copy_GBps, cus)."""
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        out = torch.zeros(4, dtype=torch.float32, device=self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def run(iters):
            e0.record()
            self._ck(self.lib.wiw_calib_mfma(self._stream(), cus, iters, out.data_ptr()), "wiw_calib_mfma")
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)
        run(1000)                                   # code load
        ms = run(20000)
        iters = max(20000, int(20000 * target_ms / max(ms, 1e-3)))
        ms = min(run(iters), run(iters))
        tf = cus * 8.0 * iters * 8 * 16384 / (ms * 1e-3) / 1e12
        # ... and what the power management lets the matrix pipe SUSTAIN: the same loop on random N(0, 0.5) operands, launches of
        # ~target_ms back to back for `sustain_s` seconds, the rate of the second half (round 6: a 6-s hold of this loop settles at
        # 1.98 PFLOP/s / 2 057 MHz / 1.31 kW where the burst above reads 2.36, profiles/r19a_mfma_energy.txt)
        sustained = None
        if sustain_s > 0:
            g = torch.Generator(device="cpu").manual_seed(7)
            src = (torch.randn(128 * 8, generator=g) * 0.5).to(self.dtype).to(self.device)
            t_end = time.perf_counter() + sustain_s
            evs = []
            while time.perf_counter() < t_end:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self._ck(self.lib.wiw_calib_mfma_random(self._stream(), cus, iters, src.data_ptr(), out.data_ptr()), "wiw_calib_mfma_random")
                b.record()
                b.synchronize()
                evs.append(a.elapsed_time(b))
            half = evs[len(evs) // 2:]
            sustained = cus * 8.0 * iters * 8 * 16384 / (sum(half) / len(half) * 1e-3) / 1e12
        src = torch.empty(1 << 28, dtype=torch.float32, device=self.device)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0.record()
        for _ in range(4):
            dst.copy_(src)
        e1.record()
        e1.synchronize()
        gbps = 4 * 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
        return {"mfma_tflops": round(tf, 1), "mfma_sustained_tflops": None if sustained is None else round(sustained, 1),
                "copy_GBps": round(gbps, 1), "cus": cus}

    def transpose(self, X, ldx, c0, rows, Cn, Y, ldy):
        """Y[c][r] = X[r][c0 + c] (bf16)."""
        self._timed("transpose", 0.0, 4.0 * rows * Cn, lambda: self._ck(
            self.lib.wiw_transpose_bf16(self._stream(), _p(X), ldx, c0, rows, Cn, _p(Y), ldy), "wiw_transpose_bf16"))
        return Y

    def adamw_step(self, p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, p16=None):
        """torch.optim.AdamW on flat fp32 tensors, in place; `p16` (this Hip's 16-bit dtype) receives the refreshed copy."""
        assert p.dtype == g.dtype == m.dtype == v.dtype == torch.float32 and p.numel() == g.numel() == m.numel() == v.numel()
        assert p16 is None or (p16.dtype == self.dtype and p16.numel() == p.numel())
        self._ck(self.lib.wiw_adamw_step(self._stream(), _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), lr, beta1, beta2,
                                         eps, weight_decay, step), "wiw_adamw_step")

    def ema_step(self, shadow, param, one_minus_decay: float):
        """EMAModel.step on flat fp32 tensors, in place: shadow -= one_minus_decay * (shadow - param)."""
        assert shadow.dtype == param.dtype == torch.float32 and shadow.numel() == param.numel()
        self._ck(self.lib.wiw_ema_step_f32(self._stream(), _p(shadow), _p(param), shadow.numel(), float(one_minus_decay)),
                 "wiw_ema_step_f32")

    def edm_loss_grad(self, pred, noisy, target, sigma):
        """EDM loss of one sample (fp32 tensors of equal shape) and its gradient w.r.t. `pred`: returns (loss 0-d tensor, grad)."""
        n = pred.numel()
        grad = torch.empty_like(pred)
        nb = min(1024, (n + 255) // 256)
        partial = torch.empty(nb, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_edm_loss_grad(self._stream(), _p(pred), _p(noisy), _p(target), n, float(sigma), _p(grad),
                                            _p(partial), nb), "wiw_edm_loss_grad")
        return partial.sum() / n, grad

    # ---- backward building blocks (row f2; csrc/train.hip)
    def colsum(self, X, rows, Cn, parts=None, units=1):
        """fp32 column sums of X [rows, Cn] (16-bit or fp32), fixed summation order.  units == 1: -> [Cn].  units > 1: X is
        `units` consecutive blocks of rows / units rows (the frames of a clip) and the result is [units, Cn], from the SAME two
        launches.  Launch 1 sums `units * k` contiguous row ranges (k: `parts`, default enough ranges to fill the chip), launch
        2 the k partials of every unit."""
        assert rows % units == 0
        ur = rows // units
        cw = 4 if X.dtype == torch.float32 else 8
        limit = max(1, 2048 // (units * (-(-Cn // (32 * cw)))))
        if parts is not None:
            limit = max(1, min(parts, ur))
        if units == 1:
            k = max(1, min(limit, ur // 64 if parts is None else limit))
        else:                                   # the ranges must not straddle units: k divides the rows of a unit
            k = next(d for d in range(min(limit, max(1, ur // 16)), 0, -1) if ur % d == 0)
        part = torch.empty(units * k, Cn, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_colsum(self._stream(), _p(X), int(X.dtype == torch.float32), rows, Cn, units * k, _p(part)), "wiw_colsum")
        if k > 1:
            out = torch.empty(units, Cn, dtype=torch.float32, device=self.device)
            self._ck(self.lib.wiw_colsum(self._stream(), _p(part), 1, units * k, Cn, units, _p(out)), "wiw_colsum")
            part = out
        return part[0] if units == 1 else part

    def layernorm_bwd(self, X, dY, gamma, rows, Cn, eps=1e-5, dres=None):
        """-> (dX 16-bit [rows, Cn], dgamma fp32 [Cn], dbeta fp32 [Cn]); dres: gradient of a residual path, added to dX."""
        nw = int(self.lib.wiw_layernorm_bwd_partials(rows))
        part = torch.empty(nw, 2 * Cn, dtype=torch.float32, device=self.device)
        dX = torch.empty(rows, Cn, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_layernorm_bwd(self._stream(), _p(X), _p(dY), _p(gamma), rows, Cn, eps, _p(dres), _p(dX), _p(part)),
                 "wiw_layernorm_bwd")
        s = self.colsum(part, nw, 2 * Cn)
        return dX, s[:Cn], s[Cn:]

    def groupnorm_bwd(self, X, dY, gamma, beta, rows, Cn, rows_per_unit, eps, silu, stats=None):
        """GroupNorm(32)(+SiLU) backward -> (dX 16-bit, dgamma fp32 [Cn], dbeta fp32 [Cn]).  stats: (mean, var) per (unit, group)
        from the forward; recomputed with wiw_groupnorm_stats when None."""
        units = rows // rows_per_unit
        if stats is None:
            rpb = self.gn_rows_per_block(rows_per_unit, False)
            stats, scratch, cnt = self._gn_buffers(rows, rows_per_unit, rpb)
            self._ck(self.lib.wiw_groupnorm_stats(self._stream(), _p(X), Cn, None, 0, rows, rows_per_unit, rpb, stats.data_ptr(),
                                                  scratch.data_ptr(), cnt.data_ptr()), "wiw_groupnorm_stats")
        # rows per block: enough (row split, unit) blocks for ~4 per CU, at least 64 rows each (the per-chunk constants are
        # loaded once per thread; the finish pass sums the splits)
        rpb_b = max(64, min(256, (rows_per_unit * units) // 1024))
        splits = -(-rows_per_unit // rpb_b)
        dX = torch.empty(rows, Cn, dtype=self.dtype, device=self.device)
        unit_cs = torch.empty(units, 2 * Cn, dtype=torch.float32, device=self.device)
        AB = torch.empty(units * 64, dtype=torch.float32, device=self.device)
        partial = torch.empty(units * splits * 2 * Cn, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_groupnorm_bwd(self._stream(), _p(X), _p(dY), _p(stats), _p(gamma), _p(beta), rows, Cn, rows_per_unit,
                                            eps, int(bool(silu)), _p(dX), _p(unit_cs), _p(AB), _p(partial), rpb_b),
                 "wiw_groupnorm_bwd")
        s = self.colsum(unit_cs, units, 2 * Cn, parts=1)
        return dX, s[Cn:], s[:Cn]

    def attn_backward(self, qkv, O, dO, seqs, S, heads, scale, Sp=None, lse=None):
        """Self-attention backward for the fused q|k|v layout: qkv [seqs*Sp, 3C], O / dO [seqs*Sp, C] (C = heads*64; sequences
        of S rows at a row stride of Sp, Sp % 16 == 0, default Sp = S) -> d_qkv [seqs*Sp, 3C] (padding rows zero).
        lse: the forward's row log-sum-exp (`attn_spatial(..., lse=...)`), spatial sequences only — skips its recomputation."""
        Sp = S if Sp is None else Sp
        Cn, M = heads * 64, seqs * Sp
        assert Sp % 16 == 0 and qkv.shape == (M, 3 * Cn) and O.shape == (M, Cn) and dO.shape == (M, Cn)
        dt, dev = self.dtype, self.device
        if lse is not None:
            assert Sp == S and S % 32 == 0 and S >= 128 and lse.numel() == seqs * heads * S
            dqkv = torch.empty(M, 3 * Cn, dtype=dt, device=dev)
            dsum = torch.empty_like(lse)
            self._ck(self.lib.wiw_attn_bwd_given_lse_bf16(self._stream(), _p(qkv), 3 * Cn, Cn, 2 * Cn, _p(O), _p(dO), Cn, _p(dqkv),
                                                          3 * Cn, _p(lse), _p(dsum), seqs, S, heads, 64, scale),
                     "wiw_attn_bwd_given_lse_bf16")
            return dqkv
        Qt = Kt = dOt = None
        if not (Sp == S and S % 32 == 0 and S >= 128):      # the one-wave-per-tile form reads transposed copies; the LDS-tiled
            Qt, Kt, dOt = (torch.empty(Cn, M, dtype=dt, device=dev) for _ in range(3))   # kernels transpose in their LDS reads
            self.transpose(qkv, 3 * Cn, 0, M, Cn, Qt, M)
            self.transpose(qkv, 3 * Cn, Cn, M, Cn, Kt, M)
            self.transpose(dO, Cn, 0, M, Cn, dOt, M)
        dqkv = torch.empty(M, 3 * Cn, dtype=dt, device=dev)
        lse = torch.empty(seqs * heads * Sp, dtype=torch.float32, device=dev)
        dsum = torch.empty_like(lse)
        self._ck(self.lib.wiw_attn_bwd_bf16(self._stream(), _p(qkv), 3 * Cn, Cn, 2 * Cn, _p(Qt), _p(Kt), _p(dOt), M, _p(O), _p(dO),
                                            Cn, _p(dqkv), 3 * Cn, _p(lse), _p(dsum), seqs, S, Sp, heads, 64, scale),
                 "wiw_attn_bwd_bf16")
        return dqkv

    def gather_taps(self, X, M, Cn, H, Wd, T=1, temporal=False, stride=1):
        """im2col rows [M, taps*Cn] of X for the conv weight gradients (9 taps, or 3 temporal ones); M = OUTPUT rows,
        (H, Wd) = output geometry (stride 2: X is the (2H, 2Wd) input)."""
        out = torch.empty(M, (3 if temporal else 9) * Cn, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_gather_taps_bf16(self._stream(), _p(X), M, Cn, H, Wd, T, int(temporal), stride, _p(out)),
                 "wiw_gather_taps_bf16")
        return out

    def wgrad_tn(self, dY, X, M, N, K, splits):
        """fp32 dW [N, K] = dY[:M]^T . X[:M] from ROW-MAJOR 16-bit operands (dY [>= M, N], X [>= M, K]); the M rows are split
        into `splits` ranges whose slabs are summed in order."""
        splits = max(1, min(int(splits), -(-M // 32)))
        while splits > 1 and (splits - 1) * (-(-(-(-M // splits)) // 32) * 32) >= M:   # every split owns at least one 32-row step
            splits -= 1
        slabs = torch.empty(splits, N * K, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_wgrad_tn_bf16(self._stream(), _p(dY), dY.stride(0), _p(X), X.stride(0), M, N, K, splits, _p(slabs)),
                 "wiw_wgrad_tn_bf16")
        if splits == 1:
            return slabs.reshape(N, K)
        out = torch.empty(1, N * K, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_colsum(self._stream(), _p(slabs), 1, splits, N * K, 1, _p(out)), "wiw_colsum")
        return out.reshape(N, K)

    def gather_taps_t(self, X, M, Cn, H, Wd, T=1, temporal=False, stride=1):
        """The im2col rows of `gather_taps`, TRANSPOSED: [taps*Cn, Mp] with Mp = M rounded up to 64 (zero columns) — the
        K-contiguous operand of the weight-gradient GEMM in one pass over X."""
        Mp = -(-M // 64) * 64
        out = torch.empty((3 if temporal else 9) * Cn, Mp, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_gather_taps_t_bf16(self._stream(), _p(X), M, Mp, Cn, H, Wd, T, int(temporal), stride, _p(out)),
                 "wiw_gather_taps_t_bf16")
        return out

    def axpby(self, X, a=1.0, Y=None, b=1.0, out=None):
        out = torch.empty_like(X) if out is None else out
        self._ck(self.lib.wiw_axpby_bf16(self._stream(), _p(X), _p(Y), a, b, X.numel(), _p(out)), "wiw_axpby_bf16")
        return out

    def silu(self, X, dY=None):
        """silu(X), or with dY: dY * silu'(X)."""
        out = torch.empty_like(X)
        self._ck(self.lib.wiw_silu_bf16(self._stream(), _p(X), _p(dY), int(dY is not None), X.numel(), _p(out)), "wiw_silu_bf16")
        return out

    def dot(self, X, Y, Z=None):
        """sum X * (Y - Z) as a 0-d fp32 tensor (fixed-order block partials)."""
        nb = min(256, (X.numel() // 8 + 255) // 256)
        part = torch.empty(nb, dtype=torch.float32, device=self.device)
        self._ck(self.lib.wiw_dot_bf16(self._stream(), _p(X), _p(Y), _p(Z), X.numel(), _p(part), nb), "wiw_dot_bf16")
        return part.sum()

    ROW_UPSAMPLE2X, ROW_DILATE2X, ROW_SUMPOOL2X2, ROW_T_TO_SEQ, ROW_SEQ_TO_T = 0, 1, 2, 3, 4

    def row_map(self, X, mode, out_rows, Cn, H=0, Wd=0, T=0, Tp=0, S=0):
        out = torch.empty(out_rows, Cn, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_row_map_bf16(self._stream(), _p(X), mode, out_rows, Cn, H, Wd, T, Tp, S, _p(out)), "wiw_row_map_bf16")
        return out

    def geglu_fwd(self, P, rows, Ch):
        H = torch.empty(rows, Ch, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_geglu_fwd(self._stream(), _p(P), rows, Ch, _p(H)), "wiw_geglu_fwd")
        return H

    def geglu_bwd(self, P, dH, rows, Ch):
        dP = torch.empty(rows, 2 * Ch, dtype=self.dtype, device=self.device)
        self._ck(self.lib.wiw_geglu_bwd(self._stream(), _p(P), _p(dH), rows, Ch, _p(dP)), "wiw_geglu_bwd")
        return dP

    def emb_combine(self, time, act, noise, Bc, B, T, E, out):
        self._ck(self.lib.wiw_emb_combine(self._stream(), _p(time), _p(act), _p(noise), Bc, B, T, E, _p(out)),
                 "wiw_emb_combine")
        return out

    def prep_unet_input(self, latents, image_latents, B, T, hw, sigma, Cpad, X):
        self._ck(self.lib.wiw_prep_unet_input(self._stream(), _p(latents), _p(image_latents), B, T, hw, sigma, Cpad,
                                              _p(X)), "wiw_prep_unet_input")
        return X

    def cfg_euler_step(self, V, ldv, latents, B, T, hw, sigma, sigma_next, gmin, gmax):
        self._ck(self.lib.wiw_cfg_euler_step(self._stream(), _p(V), ldv, _p(latents), B, T, hw, sigma, sigma_next,
                                             gmin, gmax), "wiw_cfg_euler_step")
        return latents

    def softmax_rows(self, X, ldx, rows, cols, P, ldp):
        self._ck(self.lib.wiw_softmax_rows_f32_bf16(self._stream(), _p(X), ldx, rows, cols, _p(P), ldp),
                 "wiw_softmax_rows_f32_bf16")
        return P

    def vae_time_conv_out(self, Y, ldy, weight, bias, frames, T, HW, out):
        self._ck(self.lib.wiw_vae_time_conv_out(self._stream(), _p(Y), ldy, _p(weight), _p(bias), frames, T, HW,
                                                _p(out)), "wiw_vae_time_conv_out")
        return out

    def nchw_to_nhwc(self, X, frames, Cin, HW, scale, Cpad, out):
        self._ck(self.lib.wiw_nchw_f32_to_nhwc_bf16(self._stream(), _p(X), frames, Cin, HW, scale, Cpad, _p(out)),
                 "wiw_nchw_f32_to_nhwc_bf16")
        return out
