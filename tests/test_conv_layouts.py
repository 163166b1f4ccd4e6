"""Host-side pieces of the round-4 convolution path that need no GPU: the two K orders of the 3x3 weights are permutations of
the tap-major order the reference stores (`unet.conv_k_cmajor`, `unet.conv_k_halo32`: include/wiw_svd.h WIW_K_CMAJOR /
WIW_K_HALO32), and the geometry rule of the halo-staged kernel (`Hip.conv_halo_ok`) at the served and at BASELINE config 0's
latent sizes — including the rows of ONE candidate, from which `UNetHIP._conv3` decides so that the kernel choice (and with it
the summation order) never depends on the batch in flight."""
import numpy as np
import torch

import wiw_amd  # noqa: F401
from wiw_amd.hip import Hip
from wiw_amd.unet import conv_k_cmajor, conv_k_halo32


def test_k_orders_are_permutations_of_the_tap_major_weight():
    N, C = 5, 128
    w = torch.arange(N * 9 * C, dtype=torch.float32).reshape(N, 9 * C)          # k = tap * C + c
    h, c = conv_k_halo32(w), conv_k_cmajor(w, 9)
    for n in (0, 4):
        for tap in (0, 4, 8):
            for ch in (0, 31, 32, 63, 64, 127):
                assert h[n, ((ch // 32) * 9 + tap) * 32 + ch % 32] == w[n, tap * C + ch]
                assert c[n, ((ch // 64) * 9 + tap) * 64 + ch % 64] == w[n, tap * C + ch]
    assert torch.equal(torch.sort(h, dim=1).values, torch.sort(w, dim=1).values)
    # a K tile of the halo kernel (64 consecutive k) = two taps of ONE 32-channel block, or the last tap of a block and the
    # first tap of the next
    k = np.arange(9 * C)
    blk, tap = k // (9 * 32), (k // 32) % 9
    for t0 in range(0, 9 * C, 64):
        pairs = {(int(blk[t0 + i]), int(tap[t0 + i])) for i in range(64)}
        assert len(pairs) == 2
        (b0, s0), (b1, s1) = sorted(pairs)
        assert (b0 == b1 and s1 == s0 + 1) or (b1 == b0 + 1 and s0 == 8 and s1 == 0)


def test_halo_geometry_rule():
    ok = Hip.conv_halo_ok
    T = 14
    # served 576x1024: latent 72x128, levels 72x128 / 36x64 / 18x32 / 9x16; one candidate = 2 CFG items x 14 frames
    for (H, W, N, C, expect) in [(72, 128, 320, 320, True), (36, 64, 640, 640, True), (18, 32, 1280, 1280, True),
                                 (9, 16, 1280, 1280, False)]:
        for cand in (1, 2, 3, 8):
            M = cand * 2 * T * H * W
            assert ok(M, N, C, H, W) == expect, (H, W, cand)
            assert ok(M // cand, N, C, H, W) == expect          # the per-candidate test gives the same answer for every batch
    # without CFG a candidate is 14 frames: 14 x 18 x 32 rows are 31.5 tiles — the per-candidate test says no for EVERY batch,
    # although an even number of such candidates would fill whole tiles
    assert ok(2 * T * 18 * 32, 1280, 1280, 18, 32) and not ok(T * 18 * 32, 1280, 1280, 18, 32)
    # BASELINE config 0 (256x256x8): latent 32x32 -> halo at the first level only
    assert ok(2 * 8 * 32 * 32, 320, 320, 32, 32) and not ok(2 * 8 * 16 * 16, 640, 640, 16, 16)
    # upsampling convolution: output 72x128 / 36x64 yes, 18x32 no (its staged image would be 9x16); odd heights no
    assert ok(28 * 72 * 128, 640, 640, 72, 128, up=True) and ok(28 * 36 * 64, 1280, 1280, 36, 64, up=True)
    assert not ok(28 * 18 * 32, 1280, 1280, 18, 32, up=True)
    # channel / column alignment
    assert not ok(28 * 72 * 128, 256, 320, 72, 128) and not ok(28 * 72 * 128, 320, 96, 72, 128)
