"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports exactly the symbols
declared in include/wiw_svd.h; the ctypes struct mirrors the C struct.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    import wiw_amd  # noqa: F401
    from wiw_amd.build import build

    return build(verbose=False)


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "wiw_svd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wiw_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(lib_path):
    from wiw_amd.hip import EXPORTS

    assert declared_symbols() == sorted(EXPORTS)


@pytest.mark.parametrize("variant,code", [("bf16", 0), ("fp16", 1)])
def test_library_exports_every_declared_symbol(lib_path, variant, code):
    """Both builds of the same sources (bf16: libwiwsvd.so, fp16: libwiwsvd_f16.so) export the whole header."""
    from wiw_amd.build import LIB_F16

    path = lib_path if variant == "bf16" else LIB_F16
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in wiw_svd.h but not exported by {os.path.basename(path)}"
    lib.wiw_abi_version.restype = ctypes.c_int
    lib.wiw_dtype.restype = ctypes.c_int
    assert lib.wiw_abi_version() == 17 and lib.wiw_dtype() == code


def test_gemm_args_struct_layout():
    from wiw_amd.hip import WiwGemmArgs

    # 10 pointers + 21 x 4-byte fields (+ 4 bytes of padding) + the split-K workspace pointer, natural alignment
    # (matches the C struct in wiw_svd.h)
    assert ctypes.sizeof(WiwGemmArgs) == 10 * 8 + 21 * 4 + 4 + 8 + 8 + 4 + 4     # + lnfold pointer, ln_eps, tail padding
    assert WiwGemmArgs.M.offset == 80 and WiwGemmArgs.epilogue.offset == 80 + 19 * 4
    assert WiwGemmArgs.splitk.offset == 80 + 20 * 4 and WiwGemmArgs.workspace.offset == 168
    assert WiwGemmArgs.lnfold.offset == 176 and WiwGemmArgs.ln_eps.offset == 184


def test_missing_library_fails_loudly(tmp_path):
    from wiw_amd.hip import load_library

    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        load_library(str(tmp_path / "libwiwsvd.so"))


def test_argument_validation_without_gpu(lib_path):
    """Entry points validate before launching: bad arguments return WIW_EINVAL (-1) with a message."""
    from wiw_amd.hip import WiwGemmArgs, load_library

    lib = load_library(lib_path)
    a = WiwGemmArgs()
    assert lib.wiw_gemm_bf16(None, ctypes.byref(a)) == -1
    assert b"null" in lib.wiw_last_error()
    assert lib.wiw_attn_temporal_bf16(None, 1, 192, 1, 64, 1, 17, 8, 1, 0.125) == -1
    assert b"T <= 16" in lib.wiw_last_error()
    assert lib.wiw_attn_spatial_bf16(None, 1, 128, 64, 1, 8, 1, 64, 1, 4, 1, 0.125, 1) == -1
    assert b"multiple of 8" in lib.wiw_last_error()
    assert lib.wiw_attn_spatial_ps_bf16(None, 1, 128, 64, 1, 8, 1, 60, 1, 256, 1, 1) == -1
    assert b"misaligned" in lib.wiw_last_error()
    # fused FeedForward: built for the 320 / 1280 level only; misaligned pointers are refused
    assert lib.wiw_ffn_geglu_bf16(None, 16, 640, 16, 16, 16, None, None, 0, 1, None, 0, 0.0, None, 0, 0.0, 1.0, 16, 640, 128,
                                  640, 2560, 0, 1e-5) == -1
    assert b"C = 320" in lib.wiw_last_error()
    assert lib.wiw_ffn_geglu_bf16(None, 8, 320, 16, 16, 16, None, None, 0, 1, None, 0, 0.0, None, 0, 0.0, 1.0, 16, 320, 128,
                                  320, 1280, 0, 1e-5) == -1
    assert b"16-byte aligned" in lib.wiw_last_error()
    assert lib.wiw_ffn32_geglu(None, 16, 320, 16, 16, 16, None, None, 0, 1, None, 0, 0.0, None, 0, 0.0, 1.0, 16, 320, 128, 320, 1280,
                               0, 1e-5, 9) == -1
    assert b"3-bit mask" in lib.wiw_last_error()


def test_groupnorm_onepass_geometry_rule_on_the_host(lib_path):
    """ABI 15: wiw_groupnorm_onepass_ok is a host function — the shapes of the served network's two inner levels (18 x 32 and
    9 x 16 latents, 1280 channels or a 1280 + 1280 concat) fit the one-pass kernel, every other GroupNorm of the network does not."""
    lib = ctypes.CDLL(lib_path)
    ok = lib.wiw_groupnorm_onepass_ok
    ok.restype, ok.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    frames = 28
    assert ok(1280, 0, frames * 576, 576) == 1 and ok(1280, 0, frames * 144, 144) == 1
    assert ok(1280, 1280, frames * 576, 576) == 1 and ok(1280, 1280, frames * 144, 144) == 1
    for c1, c2, hw in ((320, 0, 9216), (640, 0, 2304), (640, 320, 9216), (1280, 640, 576), (640, 0, 576), (1280, 0, 2304)):
        assert ok(c1, c2, frames * hw, hw) == 0, (c1, c2, hw)
    assert ok(1280, 0, 577, 576) == 0 and ok(1280, 0, 0, 576) == 0
