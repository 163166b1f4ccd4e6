"""HBM streaming rates of this box (torch fill / copy / read-reduce kernels): the write ceiling the GEMM epilogues see."""
import torch

dev = torch.device("cuda:0")
n = 1 << 30   # 1 GiB of bf16 pairs
x = torch.empty(n // 2, dtype=torch.bfloat16, device=dev)
y = torch.empty_like(x)
x.normal_()


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


s = t(lambda: y.fill_(1.0)); print(f"fill  (write 1 GiB): {n / s / 1e12:.2f} TB/s")
s = t(lambda: y.copy_(x)); print(f"copy  (read 1 + write 1 GiB): {2 * n / s / 1e12:.2f} TB/s total, write {n / s / 1e12:.2f}")
s = t(lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32)); print(f"sum   (read 1 GiB): {n / s / 1e12:.2f} TB/s")
z = torch.empty(n // 8, dtype=torch.bfloat16, device=dev)   # 256 MiB: fits the 256 MB MALL
s = t(lambda: z.fill_(1.0)); print(f"fill  (write 256 MiB): {n / 4 / s / 1e12:.2f} TB/s")
