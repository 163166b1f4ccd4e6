"""Where one fine-tuning step goes, by OPERATOR (not by kernel): HIP-event time of every backward closure of the tape, summed
per operator kind, plus the linear / conv backward split into dx | weight-gradient | bias pieces, and a sweep of the
weight-gradient GEMM over orientation (dW vs dW^T) and split-K factors at the served shapes.  GPU box only."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wiw_amd  # noqa: F401,E402
from wiw_amd import train as T  # noqa: E402
from wiw_amd.config import UNetConfig  # noqa: E402
from wiw_amd.hip import EPI_OUT_F32  # noqa: E402
from wiw_amd.train_unet import UNetTrain  # noqa: E402
from wiw_amd.weights import random_state_dict_torch  # noqa: E402

dev = torch.device("cuda:0")
H, W = int(os.environ.get("PH", 576)), int(os.environ.get("PW", 1024))


def step_by_operator():
    cfg = UNetConfig()
    net = UNetTrain(cfg, random_state_dict_torch(cfg, 0, dev, torch.float32), dev)
    gen = torch.Generator().manual_seed(0)
    Tn, h, w = 14, H // 8, W // 8
    lat, noise = torch.randn(1, Tn, 4, h, w, generator=gen) * 0.8, torch.randn(1, Tn, 4, h, w, generator=gen)
    st = T.prepare_step(lat, noise, 1.3, torch.randn(1, 4, h, w, generator=gen), torch.randn(1, 1, 1024, generator=gen), 0.04,
                        torch.zeros(1, Tn, 14))
    T.clear_wgrad_plans()
    T.set_wgrad_tuning(os.environ.get("TUNE", "0") == "1")
    frecs = []

    def fwrap(obj, name, label):
        f = getattr(obj, name)

        def g(*a, **k):
            if not frecs_on[0]:
                return f(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = f(*a, **k); e1.record()
            frecs.append((label, e0, e1))
            return out
        setattr(obj, name, g)
    frecs_on = [False]
    for nm in ("linear", "conv", "groupnorm", "layernorm", "geglu", "silu", "blend", "concat", "upsample2x", "cast16", "_add_rowvec",
               "host_sum"):
        fwrap(net, nm, "net." + nm)
    for nm in ("attn_spatial", "attn_small", "row_map", "transpose"):
        fwrap(net.hip, nm, "hip." + nm)
    for it in range(3):
        frecs_on[0] = it == 2
        torch.cuda.synchronize(); tf0 = time.perf_counter()
        pred = net.forward(st.unet_input, st.timestep, st.ehs, st.added_time_ids, st.action_ids)
        torch.cuda.synchronize(); tfw = time.perf_counter() - tf0
        frecs_on[0] = False
        loss, dpred = T.TrainStep(net.hip).loss_and_grad(pred, st)
        recs = []

        def wrap(f):
            def g():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record()
                recs.append((f.__qualname__, a, b))
            return g
        if it == 2:
            net.tape.ops = [wrap(f) for f in net.tape.ops]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        net.backward(dpred.reshape(pred.shape))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    agg = collections.defaultdict(lambda: [0, 0.0])
    for q, a, b in recs:
        agg[q][0] += 1; agg[q][1] += a.elapsed_time(b)
    fagg = collections.defaultdict(lambda: [0, 0.0])
    for q, e0, e1 in frecs:
        fagg[q][0] += 1; fagg[q][1] += e0.elapsed_time(e1)
    print(f"forward wall {tfw*1e3:.1f} ms (with per-call events; hip.* calls nest inside net.* ones only for attention):")
    for q, (n, ms) in sorted(fagg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {q:55s} {n:5d} calls    {ms:8.1f} ms")
    print(f"backward wall {dt*1e3:.1f} ms (with per-closure events); by operator:")
    for q, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {q:55s} {n:5d} closures {ms:8.1f} ms")
    print("plans:", {k: v for k, v in sorted(T.wgrad_plans().items())})
    T.set_wgrad_tuning(False)
    T.clear_wgrad_plans()
    del net
    torch.cuda.empty_cache()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def wgrad_sweep():
    from wiw_amd.hip import Hip
    hip = Hip(dev)
    M0 = 14 * (H // 8) * (W // 8)
    shapes = []
    for lvl, C in enumerate((320, 640, 1280)):
        M = M0 >> (2 * lvl)
        shapes += [(f"L{lvl} qkv", 3 * C, C, M), (f"L{lvl} out/proj", C, C, M), (f"L{lvl} ff1", 8 * C, C, M), (f"L{lvl} ff2", C, 4 * C, M),
                   (f"L{lvl} conv3x3", C, 9 * C, M), (f"L{lvl} conv_t", C, 3 * C, M)]
    shapes += [("L3 conv3x3", 1280, 11520, -(-(M0 >> 6) // 64) * 64), ("L3 res-t", 1280, 3840, -(-(M0 >> 6) // 64) * 64), ("up0 conv3x3 2560", 1280, 9 * 2560, M0 >> 4),
               ("up2 conv3x3 960", 320, 9 * 960, M0), ("up1 conv3x3 1920", 640, 9 * 1920, M0 >> 2), ("conv_out", 64, 2880, M0),
               ("L0 shortcut 640", 320, 640, M0)]
    print("weight-gradient GEMM dW[N,K] = dy^T x over Mrows (ms): auto split | best of sweep (orientation, split)")
    for name, N, K, M in shapes:
        dyT = torch.randn(N, M, device=dev).to(torch.bfloat16)
        xT = torch.randn(K, M, device=dev).to(torch.bfloat16)
        out = torch.empty(N, K, dtype=torch.float32, device=dev)
        outT = torch.empty(K, N, dtype=torch.float32, device=dev)
        f0, s0 = T.wgrad_plan(N, K, M)
        base = timeit(lambda: T._wgrad_run(hip, dyT, xT, N, K, M, f0, s0))
        res = []
        nk = M // 64
        cands = sorted({s for s in (1, 2, 3, 4, 6, 7, 8, 9, 12, 14, 16, 18, 21, 24, 28, 32, 36, 42, 48, 56, 63, 64, 72, 84, 96, 112, 126, 128)
                        if nk % s == 0 and nk // s >= 4})
        for flip in (0, 1):
            for s in cands:
                try:
                    if flip:
                        t = timeit(lambda: hip.gemm(xT, dyT, outT, M=K, N=N, K=M, C1=M, epilogue=EPI_OUT_F32, splitk=s), 3)
                    else:
                        t = timeit(lambda: hip.gemm(dyT, xT, out, M=N, N=K, K=M, C1=M, epilogue=EPI_OUT_F32, splitk=s), 3)
                    res.append((t, flip, s))
                except Exception as e:  # noqa: BLE001
                    res.append((1e9, flip, s))
        res.sort()
        # the row-major ("TN") kernel: no transposes; the NT numbers above exclude the two transposes it saves
        dy = dyT.t().contiguous(); x = xT.t().contiguous()
        ref = (dyT.float() @ xT.float().t())
        tn = []
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384):
            if M // sp < 128:
                break
            t = timeit(lambda: hip.wgrad_tn(dy, x, M, N, K, sp), 3)
            tn.append((t, sp))
        tn.sort()
        err = float((hip.wgrad_tn(dy, x, M, N, K, tn[0][1]) - ref).abs().max() / ref.abs().max())
        def tr2():
            a_ = torch.empty(N, M, dtype=dy.dtype, device=dev); hip.transpose(dy, N, 0, M, N, a_, M)
            b_ = torch.empty(K, M, dtype=dy.dtype, device=dev); hip.transpose(x, K, 0, M, K, b_, M)
        t_tr = timeit(tr2, 3)
        print(f"      TN best: " + ", ".join(f"s={sp} {t:.3f}" for t, sp in tn[:3]) + f"   (max err vs fp32 {err:.1e}); the two transposes NT needs: {t_tr:.3f} ms")
        del dy, x, ref
        tf = 2.0 * N * K * M / 1e9
        print(f"  {name:14s} N={N:5d} K={K:5d} M={M:6d}: plan {'T' if f0 else 'N'}/s={s0:3d} {base:7.3f} ms ({tf/base:6.0f} TF/s) | best "
              + ", ".join(f"{'T' if f else 'N'}/s={s} {t:.3f}" for t, f, s in res[:4]))
        del dyT, xT, out, outT


if __name__ == "__main__":
    if os.environ.get("ONLY", "") != "sweep":
        step_by_operator()
    if os.environ.get("ONLY", "") != "ops":
        wgrad_sweep()
