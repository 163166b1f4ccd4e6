"""Time wiw_attn_bwd_bf16 (both kernels + the three transposes of its wrapper) at the served spatial shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wiw_amd  # noqa: F401,E402
from wiw_amd.hip import Hip  # noqa: E402

dev = torch.device("cuda:0")
hip = Hip(dev)
for seqs, S, heads in ((14, 9216, 5), (14, 2304, 10), (14, 576, 20), (14, 144, 20)):
    C, M = heads * 64, seqs * S
    qkv = (torch.randn(M, 3 * C, device=dev) * 0.5).to(torch.bfloat16)
    O = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dO = torch.randn(M, C, device=dev).to(torch.bfloat16)
    for _ in range(2):
        hip.attn_backward(qkv, O, dO, seqs, S, heads, 0.125)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        hip.attn_backward(qkv, O, dO, seqs, S, heads, 0.125)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    flops = 8 * 2.0 * seqs * heads * S * S * 64          # 3 + 1 (log-sum-exp) GEMMs in dQ, 4 in dK/dV
    print(f"attn_backward seqs={seqs} S={S} heads={heads}: {ms:.3f} ms  ({flops / ms / 1e9:.0f} TFLOP/s issued)")
