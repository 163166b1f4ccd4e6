#!/bin/bash
mkdir -p gpurun_out/r03l
python - <<'PY' > gpurun_out/r03l/steps.txt 2>&1
import sys, time, torch
sys.path.insert(0, '.')
import wiw_amd
from wiw_amd import train as T
from wiw_amd.config import UNetConfig
from wiw_amd.train_unet import Trainer, UNetTrain
from wiw_amd.weights import random_state_dict_torch
dev = torch.device("cuda:0")
cfg = UNetConfig()
net = UNetTrain(cfg, random_state_dict_torch(cfg, 0, dev, torch.float32), dev)
tr = Trainer(net, lr=1e-5)
gen = torch.Generator().manual_seed(0)
Tn, h, w = 14, 72, 128
lat, noise = torch.randn(1, Tn, 4, h, w, generator=gen) * 0.8, torch.randn(1, Tn, 4, h, w, generator=gen)
aid = torch.zeros(1, Tn, 14)
st = T.prepare_step(lat, noise, 1.3, torch.randn(1, 4, h, w, generator=gen), torch.randn(1, 1, 1024, generator=gen), 0.04, aid)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred = net.forward(st.unet_input, st.timestep, st.ehs, st.added_time_ids, st.action_ids)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss, dpred = T.TrainStep(net.hip).loss_and_grad(pred, st)
    grads = net.backward(dpred.reshape(pred.shape))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for name, g in grads.items():
        net.hip.adamw_step(net.master[name].view(-1), g.reshape(-1).contiguous(), tr.m[name].view(-1), tr.v[name].view(-1), i + 1, 1e-5)
    net.refresh()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"step {i}: forward {t1-t0:.3f} s, backward {t2-t1:.3f} s, adamw+refresh {t3-t2:.3f} s; alloc {torch.cuda.memory_allocated()/2**30:.1f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB, peak {torch.cuda.max_memory_allocated()/2**30:.1f}")
    del grads, pred, dpred
PY
cat gpurun_out/r03l/steps.txt | tail -6
