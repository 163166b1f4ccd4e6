"""Same-process A/B of the weight-gradient GEMM plans (schedule model vs measured) on the full fine-tuning step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wiw_amd  # noqa: F401,E402
from wiw_amd import train as T  # noqa: E402
from wiw_amd.config import UNetConfig  # noqa: E402
from wiw_amd.train_unet import Trainer, UNetTrain  # noqa: E402
from wiw_amd.weights import random_state_dict_torch  # noqa: E402

dev = torch.device("cuda:0")
cfg = UNetConfig()
net = UNetTrain(cfg, random_state_dict_torch(cfg, 0, dev, torch.float32), dev)
tr = Trainer(net, lr=1e-5)
gen = torch.Generator().manual_seed(0)
Tn, h, w = 14, 72, 128
lat, noise = torch.randn(1, Tn, 4, h, w, generator=gen) * 0.8, torch.randn(1, Tn, 4, h, w, generator=gen)
st = T.prepare_step(lat, noise, 1.3, torch.randn(1, 4, h, w, generator=gen), torch.randn(1, 1, 1024, generator=gen), 0.04,
                    torch.zeros(1, Tn, 14))
for rnd in range(2):
    for tune in (False, True):
        T.clear_wgrad_plans()
        T.set_wgrad_tuning(tune)
        tr.step(st); tr.step(st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            tr.step(st)
        torch.cuda.synchronize()
        print(f"round {rnd} {'measured' if tune else 'model   '} plans: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per step", flush=True)
        if tune and rnd == 1:
            flips = sum(1 for v in T.wgrad_plans().values() if v[0])
            print(f"{len(T.wgrad_plans())} shapes, {flips} flipped")
            model = {k: T.wgrad_default_plan(*map(int, k.replace(',v', '').replace(',c', '').split(',')), ',c' in k) for k in T.wgrad_plans()}
            for k, v in sorted(T.wgrad_plans().items(), key=lambda kv: tuple(map(int, kv[0].replace(',v', '').replace(',c', '').split(',')))):
                print(f"  {k:22s} measured {tuple(v)} model {model[k]}")
