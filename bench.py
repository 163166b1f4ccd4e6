#!/usr/bin/env python
"""Benchmark of the SVD denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one full rollout of the hot loop: 25 Euler steps (prep_unet_input -> UNet forward ->
CFG + Euler) over one batch of synthetic candidates at 576x1024x14 (BASELINE.json configs[1]; latent
72x128, CFG batch 2 per candidate).  Inputs (cond latents, CLIP embeds, noise, actions) are already
resident in HBM when the timed region starts; weights are seeded random-init of the exact served
architecture (no checkpoint offline).  value = denoised frames/s of the whole job = N * B * 14 * K / t.

N > 1: one process per GPU (RCCL over xGMI); rank 0 owns the request of N*B candidates, SCATTERS every rank's slice
of the conditioning tensors, every rank denoises its slice, latents are GATHERED on rank 0 (weak scaling: B candidates
per GPU — `--gpus 8 --batch 8` is BASELINE config 3, 64 candidates sharded 8 per GPU).  `--total-candidates C` fixes
the request at C candidates whatever N is (strong scaling, reported as "scaling": "strong").  Both collectives are
inside the timed region; there is no exchange inside the loop.

Extra objects on the JSON line: "roofline" for the dominant kernel (HIP-event timing of every GEMM
launch inside the timed region, on the launch stream) and "cpu_baseline" (the CPU oracle timed on the
host cores on a bounded sample, rank 0 / N=1 only).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ALGO_TFLOP_PER_FORWARD = 89.604  # per UNet forward per candidate (CFG on), 576x1024x14 (reference graph)
ELIDED_TFLOP_PER_FORWARD = 3.44   # single-key cross-attentions evaluated in closed form: never credited
PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
MODE_NAMES = {0: "gemm_kernel<dense>", 1: "gemm_kernel<conv3x3>", 2: "gemm_kernel<conv3x3_s2>",
              3: "gemm_kernel<conv3x3_up>", 4: "gemm_kernel<conv_t3>"}


# the rocprofv3 kernel names behind each family (profiles/r*_kernel_stats_2step.csv rows): gemm_huge_kernel<MODE, GEGLU, SPLITK, HALO, F32E, A1>
# = the 256x320 tile (gemm_huge.hip), gemm_kernel<MODE, WAVES, STAGES, GEGLU, LNFOLD, F32E> = the 256x160 / 128x160 tiles
PROF_RES1, PROF_RES2 = 1 << 24, 1 << 25   # profile-key bits of wiw_amd.hip: the launch read a res1 / res2 operand
MODE_TEMPLATES = {
    0: ["gemm_huge_kernel<0, false, false, 0, false, true>", "gemm_huge_kernel<0, true, false, 0, false, true>",
        "gemm_huge_kernel<0, false, false, 0, false, false>", "gemm_huge_kernel<0, true, false, 0, false, false>", "gemm_kernel<0, 8, 3, false, false, false>",
        "gemm_kernel<0, 8, 3, true, false, false>", "gemm_kernel<0, 4, 2, false, false, false>",
        "gemm_kernel<0, 4, 2, true, false, false>"],
    1: ["gemm_huge_kernel<1, false, false, 1, false, false>", "gemm_huge_kernel<1, false, false, 2, false, false>", "gemm_huge_kernel<1, false, false, 0, false, false>",
        "gemm_huge_kernel<1, false, true, 0, false, false>", "gemm_kernel<1, 8, 3, false, false, false>",
        "gemm_kernel<1, 4, 2, false, false, false>", "splitk_reduce_kernel"],
    2: ["gemm_huge_kernel<2, false, false, 0, false, false>", "gemm_kernel<2, 8, 3, false, false, false>", "gemm_kernel<2, 4, 2, false, false, false>"],
    3: ["gemm_huge_kernel<3, false, false, 1, false, false>", "gemm_huge_kernel<3, false, false, 0, false, false>", "gemm_kernel<3, 8, 3, false, false, false>", "gemm_kernel<3, 4, 2, false, false, false>"],
    4: ["gemm_huge_kernel<4, false, false, 0, false, false>", "gemm_kernel<4, 8, 3, false, false, false>", "gemm_kernel<4, 4, 2, false, false, false>"],
}


def synth_actions(B, T):
    base = [4] + [[1, 2, 1, 3][i % 4] for i in range(T - 1)]  # exercises both rolls
    return np.tile(np.array(base, dtype=np.int64), (B, 1))


def _cpu_model() -> str:
    """Host CPU model string (SURVEY 8d asks for it next to the core count)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or "unknown"


def cpu_baseline(sd_cpu, cfg, threads, device=None, dtype=None):
    """Oracle ("port") on the host cores: ONE fp32 UNet forward of BASELINE config 0 (256x256x8, CFG on)
    = 1/10 of its 10-step rollout; frames/s extrapolated as 8 / (10 * t_forward).  With `device` given the HIP path
    evaluates the SAME forward (same weights, same inputs) and the line carries its deviation from the oracle: the
    bench number is then self-checking (the oracle stays the checker, never the thing measured)."""
    import svd_oracle as O

    torch.set_num_threads(threads)
    T, h, w = 8, 32, 32
    ocfg = dict(cfg.as_dict(), num_frames=T, action_input_channel=cfg.action_input_channel)
    rs = np.random.RandomState(0)
    sample = torch.from_numpy(rs.standard_normal((2, T, 8, h, w)).astype(np.float32))
    ehs = torch.from_numpy(rs.standard_normal((2, 1, cfg.cross_attention_dim)).astype(np.float32))
    ehs[:1] = 0            # CFG convention of the pipeline (pipeline:221-227, 244-250): the unconditional half carries
    sample[:1, :, 4:] = 0  # zero image embeddings and zero conditioning latents
    tids = torch.tensor([[6, 127, 0.02]] * 2)
    # the nav checkpoint embeds 14-channel action rows; the 8-frame config uses the first 8 frames' rows
    aid = torch.from_numpy(O.action_ids_idx_encode(synth_actions(1, cfg.action_input_channel)))[:, :T]
    t0 = time.time()
    with torch.no_grad():
        ref = O.unet_forward(sd_cpu, ocfg, sample, torch.tensor(1.0), ehs, tids, aid)
    dt = time.time() - t0
    res = {"value": round(8.0 / (10.0 * dt), 5), "unit": "frames/s", "cores": threads, "kind": "port",
           "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(), "seconds_per_forward": round(dt, 3),
           "sample": "1 of the 10 UNet forwards (CFG batch 2, fp32, full-size weights) of BASELINE config 0 "
                     "(256x256x8, 10 steps); frames/s = 8 / (10 * t_forward)"}
    if device is not None:
        from wiw_amd.config import UNetConfig
        from wiw_amd.unet import UNetHIP

        cfg8 = UNetConfig(num_frames=T, action_input_channel=cfg.action_input_channel)
        u8 = UNetHIP(cfg8, sd_cpu, device, dtype=dtype or torch.bfloat16)
        out = u8(sample, 1.0, ehs, tids, aid).float().cpu()
        err = out - ref
        res["hip_vs_oracle"] = {
            "rms_rel": round(float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), 6),
            "max_rel": round(float(err.abs().max() / ref.abs().max()), 6),
            "what": "the HIP path on the same forward (same weights and inputs, 16-bit storage) against the fp32 oracle"}
        del u8
        torch.cuda.empty_cache()
    return res


class PowerSampler:
    """Shader clock and package power of GPU 0 from rocm-smi (a few samples per second, a thread of this process) over the
    timed region.  Why it is in the line: the denoising loop runs against the board's power management — every MFMA-heavy
    kernel of this repo holds ~1.3 kW at 1.7-2.2 GHz of the 2.4 GHz the peak figures assume (profiles/r16k_power_by_kernel.txt)
    — so frames/s on a box follows its clock.  None when rocm-smi is not there."""

    PERIOD_S = 0.4

    def __init__(self):
        self.samples, self._stop, self._th = [], False, None

    @staticmethod
    def read():
        import re
        import subprocess
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
            r = [x for x in out.splitlines() if x.startswith("card")][0]
            clocks = [int(x) for x in re.findall(r"\((\d+)Mhz\)", r)]
            return max(clocks[2:4]), float(r.split(",")[-1])
        except Exception:
            return None

    def start(self):
        import threading

        def loop():
            import time as _t
            while not self._stop:
                smp = self.read()
                if smp is None:
                    return
                self.samples.append(smp)
                _t.sleep(self.PERIOD_S)        # no back-to-back forks of rocm-smi beside the timed launches

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=15)
        sm = self.samples[1:] if len(self.samples) > 2 else self.samples      # the first call may predate the first launch
        if not sm:
            return None
        return {"sclk_MHz_mean": round(sum(x[0] for x in sm) / len(sm)), "sclk_MHz_min": min(x[0] for x in sm),
                "watts_mean": round(sum(x[1] for x in sm) / len(sm)), "watts_max": round(max(x[1] for x in sm)), "samples": len(sm),
                "sampling": f"on (rank 0 only): one rocm-smi fork every {self.PERIOD_S} s + its own run time, from a thread of this process",
                "source": "rocm-smi --showclocks --showpower during the timed region (GPU 0); nominal peak clock 2400 MHz, board limit 1400 W"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed rollouts (each = num-inference-steps Euler steps)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="candidates per GPU (weak scaling)")
    ap.add_argument("--total-candidates", type=int, default=0,
                    help="strong scaling: the request holds this many candidates in total, sharded over the N ranks")
    ap.add_argument("--num-inference-steps", type=int, default=25)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default="bf16",
                    help="16-bit storage / MFMA operand type: bf16 (BASELINE's, libwiwsvd.so) or fp16 (the reference's "
                         "served default, libwiwsvd_f16.so)")
    ap.add_argument("--residual-fp32-full", action="store_true", help="UNetHIP(residual_fp32='full'): rounds 4-5's form of the mode")
    ap.add_argument("--residual-fp32", action="store_true",
                    help="UNetHIP(residual_fp32=True): the residual stream in fp32 (with --dtype fp16 the configuration within "
                         "1e-3 of the reference's fp32 evaluation, DESIGN.md 5)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extra` legs the default single-GPU line appends after its timed region (B = 8 rollout = "
                         "BASELINE config 2's per-GPU work, fp16 and fp16 + fp32-residual rollouts, a --train leg of 2 warm-up + 5 timed steps)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="skip the box calibration launches (rocprofv3 runs: ~100 ms of calibration kernels would lead the "
                         "kernel statistics); `box` and `roofline.frac_of_box_peak` are then absent")
    ap.add_argument("--tiny", action="store_true", help="reduced-width model (plumbing check only; INVALID as a result)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample rocm-smi (clock, watts) during the timed region")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every UNet forward launch by launch (default: replay a captured hipGraph on the Euler steps "
                         "that carry no per-launch events; same bytes)")
    ap.add_argument("--event-every", type=int, default=5,
                    help="per-launch HIP events are recorded on every n-th Euler step of the timed rollouts (every launch of "
                         "those steps); n = 1 instruments every step and costs 3.4 %% of frames/s (21 k extra events / rollout)")
    ap.add_argument("--dump-shapes", type=str, default="", help="write per-(M,N,K) GEMM timings to this file")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4 instead of the inference loop: one fine-tuning step per 'step' (un-fused training "
                         "forward, EDM loss, backward through every operator, AdamW), one sample per GPU, data parallel over "
                         "the ranks (ShardedAdamW: gradient reduce-scatter overlapped with the backward + parameter all-gather).")
    ap.add_argument("--no-autotune", action="store_true", help="--train: the schedule model's weight-gradient GEMM plans instead "
                    "of the ones measured during warm-up")
    ap.add_argument("--train-height", type=int, default=576, help="train_svd.sh:22-23 trains at 576x1024 (the dataset comments "
                    "of train_svd.py:846-855 show 256x512 clips: pass 256 / 512 for those)")
    ap.add_argument("--train-width", type=int, default=1024)
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time whole requests (uint8 panorama in -> uint8 frames out: CLIP + VAE encode, denoise, VAE "
                         "decode, PIL post-processing) through server.worker.SVDWorker; reported as 'end_to_end'")
    args = ap.parse_args()

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    # WIW_FORCE_DIST=1: take the multi-rank code path (env-driven RCCL process group, scatter / gather of candidate slices,
    # ZeRO-1 optimiser) even with ONE rank — what `torchrun --nproc-per-node 1 bench.py --gpus 1` exercises on a one-GPU box
    dist_on = world > 1 or bool(os.environ.get("WIW_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)     # RANK / WORLD_SIZE / MASTER_* from the launcher's environment

    import wiw_amd  # noqa: F401
    if args.train:
        line = train_bench(args, rank, world, device, dist_on)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    from wiw_amd.config import UNetConfig
    from wiw_amd.parallel import sharded_denoise
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict_torch

    cfg = UNetConfig.tiny(14) if args.tiny else UNetConfig()
    T = cfg.num_frames
    h, w = args.height // 8, args.width // 8
    sd = random_state_dict_torch(cfg, 0, device, torch.float32)
    unet = UNetHIP(cfg, sd, device, dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16,
                   residual_fp32="full" if args.residual_fp32_full else (True if args.residual_fp32 else None))
    # what THIS box gives a pure-MFMA loop and a device copy (outside the timed region): the pool's boxes differ by +-5 %
    box = unet.hip.calibrate_box() if rank == 0 and not args.no_calibrate else None
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    torch.cuda.empty_cache()
    den = SVDDenoiser(unet, use_graph=not args.no_graph)

    B = args.batch
    Btot = B * world
    strong = args.total_candidates > 0
    if strong:
        Btot = args.total_candidates
        B = -(-Btot // world)          # largest slice (rank 0's)
    g = torch.Generator(device="cpu").manual_seed(1234)
    req = None
    if rank == 0:
        req = dict(image_latents=torch.randn(Btot, 4, h, w, generator=g).to(device),
                   image_embeddings=torch.randn(Btot, 1, cfg.cross_attention_dim, generator=g).to(device),
                   noise=torch.randn(Btot, T, 4, h, w, generator=torch.Generator().manual_seed(1)).to(device),
                   actions=synth_actions(Btot, T))

    # Per-launch HIP events (on the launch stream) INSIDE the timed region, on a sample of its Euler steps: every step
    # issues the same launches, so per-kernel averages from every n-th step are the rollout's; the other steps run
    # without the event packets (two per launch: measured 3.4 % of frames/s when every step carries them).
    ev = {"on": False, "gemm": [], "kern": [], "steps": 0}
    every = max(1, args.event_every)

    def arm(step_index):
        on = ev["on"] and step_index % every == 0 and step_index < args.num_inference_steps
        unet.hip.gemm_profile = ev["gemm"] if on else None
        unet.hip.kernel_profile = ev["kern"] if on else None
        ev["steps"] += int(on)

    def after_step(i, _lat):
        arm(i + 1)

    def rollout():
        arm(0)
        if not dist_on:
            out_ = den.denoise(req["image_latents"], req["image_embeddings"], req["noise"], req["actions"],
                               num_steps=args.num_inference_steps, callback=after_step)
        else:
            r = req or dict(image_latents=None, image_embeddings=None, noise=None, actions=None)
            out_ = sharded_denoise(den.denoise, device, r["image_latents"], r["image_embeddings"], r["noise"], r["actions"],
                                   num_steps=args.num_inference_steps, callback=after_step)
        arm(args.num_inference_steps)     # off
        return out_

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        rollout()
    ev["on"] = not args.no_kernel_events
    den.host_launch = {"eager": [0.0, 0], "graph": [0.0, 0]}
    barrier()
    pws = PowerSampler() if rank == 0 and not args.no_power else None
    if pws is not None:
        pws.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = rollout()
    barrier()
    dt = time.perf_counter() - t0
    power = pws.stop() if pws is not None else None
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    prof, kprof = ev["gemm"], ev["kern"]
    unet.hip.gemm_profile = None
    unet.hip.kernel_profile = None
    # fraction of the timed Euler steps that carried events: per-family seconds / FLOPs below are scaled by 1 / ev_frac
    ev_frac = ev["steps"] / float(args.num_inference_steps * args.steps) if ev["steps"] else 0.0
    if rank == 0:
        assert out is not None and torch.isfinite(out).all(), "non-finite latents"
        frames = Btot * T * args.steps
        value = frames / dt
        res = {
            "metric": "denoised frames/sec (576x1024x14, 25 steps)", "value": round(value, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            **({"residual_stream": "fp32"} if unet.res32_any else {}),
            "config": {"workload": f"SVD denoise loop {args.height}x{args.width}x{T}, {args.num_inference_steps} Euler steps, "
                                   f"CFG on, " + (f"{Btot} candidates in total" if strong else f"{B} candidate(s)/GPU") + ", random-init weights" + (" [TINY MODEL - INVALID]" if args.tiny else ""),
                       "candidates_per_gpu": B, "parallelism": f"candidate-sharded x{world}",
                       "collectives": f"rccl process group, {world} rank(s): scatter / gather of candidate slices" if dist_on else "none (single process)"},
        }
        # host time spent enqueueing one UNet forward (~1 100 launches through ctypes, or one hipGraphLaunch): measured around the
        # calls of the timed region, no synchronisation inside; the event-carrying steps are the eager sample
        if box is not None:
            res["box"] = {"mfma_peak_measured": box["mfma_tflops"], "mfma_sustained": box.get("mfma_sustained_tflops"),
                          "mfma_peak_unit": "TFLOP/s", "hbm_copy_GBps": box["copy_GBps"],
                          "cus": box["cus"], "what": "a ~50 ms launch of back-to-back MFMAs on (near-constant) register operands, 8 "
                          "waves per CU; mfma_sustained: the same loop on random N(0, 0.5) operands held for 1 s, rate of the second "
                          "half = what the board's power management lets the matrix pipe sustain; a 1 GiB device copy (read + write "
                          "bytes); all measured on this box before the timed region"}
        if power is not None:
            res["power"] = power
        hl = den.host_launch
        res["host_launch"] = {
            "hip_graph": (not args.no_graph) and den.graph_error is None,
            **({"graph_error": den.graph_error} if den.graph_error else {}),
            "eager_ms_per_forward": round(1e3 * hl["eager"][0] / hl["eager"][1], 3) if hl["eager"][1] else None,
            "graph_replay_ms_per_forward": round(1e3 * hl["graph"][0] / hl["graph"][1], 3) if hl["graph"][1] else None,
            "forwards": {"eager": hl["eager"][1], "graph": hl["graph"][1]},
            "note": "Euler steps that carry per-launch HIP events (every --event-every-th) run EAGERLY on the graph's static "
                    "buffers inside the timed region; the others are graph replays"}
        fwd_per_step = args.num_inference_steps * (Btot / world if strong else B)   # candidate-forwards per GPU
        algo = ALGO_TFLOP_PER_FORWARD * (h * w) / (72 * 128) * fwd_per_step  # linear in pixels & candidates
        if not args.tiny:
            # utilisation is quoted from FLOPs ISSUED to the matrix pipe (sum over the timed launches of 2*M*N*K per
            # GEMM / conv launch + the attention kernels' Q.K^T / P.V); without per-launch events: the reference graph's
            # count minus the exactly-elided cross-attention.  `algorithmic_tflop_per_step` (the reference graph, 89.604
            # TF / forward) is reported for the frames/s <-> FLOP conversion only and is NOT a utilisation numerator.
            res["algorithmic_tflop_per_step"] = round(algo, 1)
            issued = (ALGO_TFLOP_PER_FORWARD - ELIDED_TFLOP_PER_FORWARD) * (h * w) / (72 * 128) * fwd_per_step * args.steps
            src = "reference graph minus elided cross-attention (no per-launch events)"
            if prof:
                issued = (sum(e[2] for e in prof) + sum(e[3] for e in (kprof or []))) / 1e12 / ev_frac
                src = "sum of per-launch algorithmic FLOPs of the timed region (GEMM/conv launches + attention kernels)"
                res["kernel_events"] = (f"HIP events around every launch of every {every}-th Euler step of the timed rollouts "
                                        f"({ev['steps']} of {args.num_inference_steps * args.steps} steps); seconds / launches "
                                        "below are of those steps, shares are relative to their part of the timed region")
            res["issued_tflop_per_step"] = round(issued / args.steps, 1)
            res["mfma_util"] = round(issued / dt / PEAK_BF16_TFLOPS, 4)
            res["mfma_util_source"] = src
        if prof:
            by_mode = {}
            shapes = {}
            for e0, e1, fl, mode, shp in prof:
                d = by_mode.setdefault(mode, [0.0, 0.0, 0])
                d[0] += e0.elapsed_time(e1) * 1e-3
                d[1] += fl
                d[2] += 1
                sh = shapes.setdefault((mode,) + shp, [0.0, 0])
                sh[0] += e0.elapsed_time(e1) * 1e-3
                sh[1] += 1
            dom = max(by_mode, key=lambda m: by_mode[m][0])
            traffic, traffic_src = pmc_traffic(dom)
            tsec, fl, cnt = by_mode[dom]
            ach = fl / tsec / 1e12
            res["roofline"] = {"kernel": MODE_NAMES[dom], "templates": MODE_TEMPLATES.get(dom, []), "bound": "mfma",
                               "achieved": round(ach, 1),
                               "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               **({"frac_of_box_peak": round(ach / box["mfma_tflops"], 4)} if box is not None else {}),
                               **({"frac_of_box_sustained": round(ach / box["mfma_sustained_tflops"], 4)}
                                  if box is not None and box.get("mfma_sustained_tflops") else {}),
                               # the board's power management holds the loop below the 2.4 GHz the peak assumes:
                               # the same fraction against the peak at the mean clock of THIS timed region
                               **({"frac_at_measured_clock": round(ach / (PEAK_BF16_TFLOPS * power["sclk_MHz_mean"] / 2400.0), 4)}
                                  if power is not None and power.get("sclk_MHz_mean") else {}),
                               "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                               "launches": cnt, "avg_launch_us": round(1e6 * tsec / cnt, 1),
                               "share_of_timed_region": round(tsec / (dt * ev_frac), 3)}
            res["gemm_kernels"] = {MODE_NAMES[m]: {"launches": v[2], "seconds": round(v[0], 4),
                                                   "tflops": round(v[1] / v[0] / 1e12, 1)} for m, v in sorted(by_mode.items())}
            # the dense family mixes regimes: the K <= 320 launches (C = 320 level) sit below the machine
            # ridge — 70 % of their time is the GEGLU up-projection, whose exact-erf epilogue is VALU-bound, the rest
            # streams at 4.0-4.4 TB/s; algorithmic bytes = 16-bit A + W + out + the residual operands the launch reads (round 6:
            # rounds 1-5 left the residuals out and under-stated these launches by a quarter — profiles/r19i_thin320_probe.txt)
            lo = [0.0, 0.0, 0.0, 0]
            hi = [0.0, 0.0, 0]
            for (mode, M, N, K, epi), (sec, cnt) in shapes.items():
                if mode != 0:
                    continue
                n_out = N // 2 if epi & 1 else N
                if K <= 320:
                    res_b = sum((4.0 if epi & f32 else 2.0) * M * n_out for has, f32 in ((PROF_RES1, 128), (PROF_RES2, 256)) if epi & has)
                    out_b = (4.0 if epi & 4 else 2.0) * M * n_out
                    lo[0] += sec; lo[1] += 2.0 * M * N * K * cnt; lo[2] += (2.0 * (M * K + N * K) + out_b + res_b) * cnt; lo[3] += cnt
                else:
                    hi[0] += sec; hi[1] += 2.0 * M * N * K * cnt; hi[2] += cnt
            if lo[0] > 0 and hi[0] > 0:
                res["dense_split"] = {
                    "K<=320": {"launches": lo[3], "seconds": round(lo[0], 4), "tflops": round(lo[1] / lo[0] / 1e12, 1),
                               "algorithmic_GBps": round(lo[2] / lo[0] / 1e9, 1), "peak_GBps": 8000.0},
                    "K>320": {"launches": hi[2], "seconds": round(hi[0], 4), "tflops": round(hi[1] / hi[0] / 1e12, 1)}}
            if kprof:
                fam = {}
                for e0, e1, name, fl, nb_ in kprof:
                    d = fam.setdefault(name, [0.0, 0.0, 0.0, 0])
                    d[0] += e0.elapsed_time(e1) * 1e-3
                    d[1] += fl
                    d[2] += nb_
                    d[3] += 1
                res["other_kernels"] = {
                    k: dict(launches=v[3], seconds=round(v[0], 4), share_of_timed_region=round(v[0] / (dt * ev_frac), 4),
                            **({"tflops": round(v[1] / v[0] / 1e12, 1)} if v[1] > 0 else {}),
                            algorithmic_GBps=round(v[2] / v[0] / 1e9, 1))
                    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
            if args.dump_shapes:
                rows = sorted(((k, v) for k, v in shapes.items()), key=lambda kv: -kv[1][0])
                with open(args.dump_shapes, "w") as f:
                    for (mode, M, N, K, epi), (sec, cnt) in rows:
                        n_out = N // 2 if epi & 1 else N
                        nres = sum((2 if epi & f32 else 1) for has, f32 in ((PROF_RES1, 128), (PROF_RES2, 256)) if epi & has)
                        # algorithmic bytes: the A operand once (not per tap), W, the output, the residual operands read (round 6)
                        mb = 2.0 * (M * K / (9 if mode in (1, 2, 3, 5) else (3 if mode == 4 else 1)) + N * K
                                    + M * n_out * ((2 if epi & 4 else 1) + nres)) / 1e6
                        f.write(f"mode={mode} M={M} N={N} K={K} epi={epi & 0xFFFFFF} res={bool(epi & PROF_RES1) + bool(epi & PROF_RES2)} "
                                f"launches={cnt} total_ms={1e3 * sec:.2f} avg_us={1e6 * sec / cnt:.1f} "
                                f"tflops={2.0 * M * N * K * cnt / sec / 1e12:.1f} algorithmic_MB={mb:.1f} "
                                f"algorithmic_GBps={mb * cnt / sec / 1e3:.0f}\n")
                    for k, v in (res.get("other_kernels") or {}).items():
                        f.write(f"family={k} " + " ".join(f"{a}={b}" for a, b in v.items()) + "\n")
        if args.end_to_end and world == 1 and not args.tiny:
            res["end_to_end"] = end_to_end(den, unet, device, B, args)
        if (world == 1 and not dist_on and not args.no_extras and not args.tiny and B == 1 and not strong and
                (args.height, args.width, args.num_inference_steps) == (576, 1024, 25) and not (args.residual_fp32 or args.residual_fp32_full)):
            res["extra"] = extras(args, cfg, unet, device, req, time.perf_counter() - t_start)
        if sd_cpu is not None:
            try:
                res["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, min(os.cpu_count() or 1, 32), device, unet.dtype)
            except Exception as e:  # the GPU number stands on its own; report why the baseline is absent
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def train_bench(args, rank, world, device, dist_on=False):
    """BASELINE config 4: FTsvd/train_svd.py step (one sample per GPU and micro-batch, train_svd.sh:26; 576x1024x14 as
    train_svd.sh:22-24) on the served architecture with random-init weights and a synthetic latent batch.
    metric = training samples / s (one micro-batch per step, the optimiser on every one)."""
    import torch.distributed as dist

    from wiw_amd import train as T
    from wiw_amd.config import UNetConfig
    from wiw_amd.parallel import ShardedAdamW
    from wiw_amd.train_unet import Trainer, UNetTrain
    from wiw_amd.weights import random_state_dict_torch

    cfg = UNetConfig.tiny(14) if args.tiny else UNetConfig()
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    net = UNetTrain(cfg, random_state_dict_torch(cfg, 0, device, torch.float32), device, dtype=dtype)
    opt = None
    if dist_on:
        opt = ShardedAdamW(Trainer.optimizer_shapes(net), device,   # trainable AND live parameters only
                           lambda p, g, m, v, step, lr, b1, b2, eps, wd: net.hip.adamw_step(p, g, m, v, step, lr, b1, b2, eps, wd),
                           lr=1e-5)
    tr = Trainer(net, lr=1e-5, optimizer=opt, autotune=not args.no_autotune)
    h, w, Tn = args.train_height // 8, args.train_width // 8, cfg.num_frames
    gen = torch.Generator().manual_seed(100 + rank)                       # every rank its own sample
    lat, noise = torch.randn(1, Tn, 4, h, w, generator=gen) * 0.8, torch.randn(1, Tn, 4, h, w, generator=gen)
    aid = torch.zeros(1, Tn, cfg.action_input_channel)
    seq = [4] + [[1, 2, 1, 3][i % 4] for i in range(Tn - 1)]
    for i in range(Tn):
        aid[0, i, : i + 1] = torch.tensor(seq[: i + 1], dtype=torch.float32)   # utils/svd_utils.py:594-632
    st = T.prepare_step(lat, noise, 1.3, torch.randn(1, 4, h, w, generator=gen),
                        torch.randn(1, 1, cfg.cross_attention_dim, generator=gen), 0.04, aid)

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        tr.step(st)
    barrier()
    torch.cuda.reset_peak_memory_stats(device)
    pws = PowerSampler() if rank == 0 and not getattr(args, "no_power", False) else None
    if pws is not None:
        pws.start()
    # every step is bracketed (a step is ~0.3 s of GPU work: the barrier costs nothing measurable) so that the line can say
    # whether a slow figure is one outlier (allocator growth, first touch) or the steady state — VERDICT r5 item 3
    per_step = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        loss = tr.step(st)
        barrier()
        per_step.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    power = pws.stop() if pws is not None else None
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist_on:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    line = None
    if rank == 0:
        assert np.isfinite(loss)
        line = ({
            "metric": "fine-tuning samples/sec (train_svd.py step: fwd + bwd + AdamW)", "value": round(world * args.steps / dt, 4),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "ms_per_step_median": round(1e3 * float(np.median(per_step)), 2), "ms_per_step_max": round(1e3 * max(per_step), 2),
            "ms_per_step_all": [round(1e3 * x, 1) for x in per_step], "power": power,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"SVD UNet fine-tuning step {args.train_height}x{args.train_width}x{Tn}, one sample per GPU, "
                                   "random-init weights; un-fused training forward, GEMM-shaped gradients on the inference GEMM "
                                   "kernel (orientation / split-K per shape " + ("from the schedule model" if args.no_autotune else
                                   "measured during warm-up") + "), LDS-tiled attention backward" +
                                   (" [TINY MODEL - INVALID]" if args.tiny else ""),
                       "parallelism": f"data-parallel x{world}, ZeRO-1 (reduce-scatter + all-gather)" if dist_on else "single GPU"},
            "final_loss": round(float(loss), 5), "peak_memory_GiB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1)})
    del tr, net, opt
    torch.cuda.empty_cache()
    return line


def extras(args, cfg, unet, device, req, elapsed):
    """Legs appended to the default single-GPU line AFTER its timed region: numbers the driver's clock
    otherwise never sees.  Each leg is guarded; none touches `value`.
      batch8      one 25-step rollout of 8 candidates on this GPU = the per-GPU work of BASELINE configs 2 / 3
      fp16        one rollout with the fp16 library (the reference's served dtype, eval_inference.py:294)
      fp16_res32  the same with the BLOCK-LEVEL residual stream in fp32 (UNetHIP(residual_fp32=True), round 6): the configuration whose
                  25-step latents at this size are gated at <= 1e-3 (rms AND max) against the reference pipeline's; `ratio_to_fp16`
      fp16_res32_full   ... with the hidden stream inside the transformer blocks in fp32 too (residual_fp32="full", rounds 4-5)
      end_to_end  ONE whole request through the worker: CLIP + VAE encode, loop, temporal VAE decode (own
                  row with its TFLOP/s), PIL resize, uint8 response
      train       2 warm-up + 5 timed fine-tuning steps at 576x1024x14 (BASELINE config 4's per-GPU work), every step bracketed:
                  mean / median / max, clock and watts"""
    import copy

    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict_torch

    out = {}
    T, h, w = cfg.num_frames, args.height // 8, args.width // 8
    budget = 270.0     # seconds of bench wall clock after which no further leg starts

    def timed_rollout(den, r, n):
        den.denoise(r["image_latents"], r["image_embeddings"], r["noise"], r["actions"], num_steps=2)    # warm (capture) pass
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        lat = den.denoise(r["image_latents"], r["image_embeddings"], r["noise"], r["actions"], num_steps=args.num_inference_steps)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        assert torch.isfinite(lat).all()
        return {"frames_per_s": round(n * T / dt, 4), "seconds_per_rollout": round(dt, 3), "candidates": n}

    t0 = time.perf_counter()

    def left():
        return budget - elapsed - (time.perf_counter() - t0)

    try:
        if left() > 60:
            g = torch.Generator(device="cpu").manual_seed(4321)
            r8 = dict(image_latents=torch.randn(8, 4, h, w, generator=g).to(device),
                      image_embeddings=torch.randn(8, 1, cfg.cross_attention_dim, generator=g).to(device),
                      noise=torch.randn(8, T, 4, h, w, generator=g).to(device), actions=synth_actions(8, T))
            out["batch8"] = timed_rollout(SVDDenoiser(unet, use_graph=False), r8, 8)
            del r8
            torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        out["batch8"] = {"error": f"{type(e).__name__}: {e}"}
    if args.dtype == "bf16":
        for name, r32 in (("fp16", False), ("fp16_res32", True), ("fp16_res32_full", "full")):
            try:
                if left() > 30:
                    u = UNetHIP(cfg, random_state_dict_torch(cfg, 0, device, torch.float32), device, dtype=torch.float16,
                                residual_fp32=r32)
                    out[name] = timed_rollout(SVDDenoiser(u, use_graph=True), req, 1)
                    if name != "fp16" and "frames_per_s" in out.get("fp16", {}):
                        out[name]["ratio_to_fp16"] = round(out[name]["frames_per_s"] / out["fp16"]["frames_per_s"], 4)
                    del u
                    torch.cuda.empty_cache()
            except Exception as e:   # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"}
    try:      # one whole request through the worker (CLIP + VAE encode, loop, VAE decode, PIL, uint8), with a vae_decode row
        if left() > 40:
            out["end_to_end"] = end_to_end(SVDDenoiser(unet, use_graph=True), unet, device, 1, args)
            torch.cuda.empty_cache()
    except Exception as e:   # noqa: BLE001
        out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        if left() > 45:
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.no_autotune = 5, 2, True
            gc.collect()
            torch.cuda.empty_cache()           # four legs have churned the allocator: start the step from a clean cache
            tl = train_bench(a2, 0, 1, device, False)
            out["train"] = {k: tl[k] for k in ("metric", "value", "unit", "ms_per_step", "ms_per_step_median", "ms_per_step_max",
                                               "ms_per_step_all", "steps", "warmup", "power", "peak_memory_GiB")}
    except Exception as e:   # noqa: BLE001
        out["train"] = {"error": f"{type(e).__name__}: {e}"}
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def end_to_end(den, unet, device, B, args):
    """Whole requests through the worker ('also report end-to-end (incl. CLIP/VAE) separately'):
    random-init VAE / CLIP of the production geometry, B candidates, args.num_inference_steps Euler steps."""
    import numpy as np
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from wiw_amd import frontend as FE
    from wiw_amd.server.worker import SVDWorker
    from wiw_amd.vae import HIPFrontend, VAEHIP

    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(
        hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
        patch_size=14, projection_dim=1024, hidden_act="gelu")).eval()
    fe = HIPFrontend(VAEHIP(FE.vae_random_state_dict(1), device, hip=unet.hip), clip, device_io=True)

    def denoise(il, ie, nz, act, **kw):   # as serve_worker.build_worker: latents stay on the device
        return den.denoise(torch.as_tensor(il), torch.as_tensor(ie), torch.as_tensor(nz), act, **kw)

    worker = SVDWorker(denoise, fe, width=args.width, height=args.height, num_inference_steps=args.num_inference_steps)
    rs = np.random.RandomState(0)
    req = {"b_action": np.asarray(synth_actions(B, 14), dtype=np.int64), "save_dirs": [f"/tmp/wiw_e2e_{i}" for i in range(B)],
           "request_model_name": "igen", "b_image": rs.randint(0, 256, size=(B, 3, args.height, args.width), dtype=np.uint8),
           "return_objects": [True] * B}
    worker(req)                      # warm-up (CLIP / hipBLASLt first-call costs)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = worker(req)
    dt = time.perf_counter() - t0
    assert out["pred_frames"].shape == (B, 14, 3, 480, 480)
    # the phases either side of the loop, HIP events on the current stream (row f1 / f4 of SURVEY 8: temporal VAE decoder
    # autoencoder_kl_temporal_decoder.py:87-161; CLIP + VAE encoder pipeline:183-252)
    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize(device)
        return r, e0.elapsed_time(e1) * 1e-3
    x = rs.standard_normal((B, 3, args.height, args.width)).astype(np.float32)
    nz = rs.standard_normal(x.shape).astype(np.float32)
    _, t_enc = timed(lambda: fe.encode(x, nz, 0.02))
    lat = torch.randn(B, 14, 4, args.height // 8, args.width // 8, device=device)
    _, t_dec = timed(lambda: fe.decode_uint8(lat))
    dec_tf = 97.2e12 * B * (args.height * args.width) / (576.0 * 1024.0)    # 97.2 TFLOP per 576x1024x14 clip (temporal decoder MACs x 2)
    return {"seconds_per_request": round(dt, 3), "frames_per_s": round(B * 14 / dt, 3), "candidates": B,
            "includes": "CLIP + VAE encode, denoise loop, VAE decode, PIL resize, uint8 response (random-init weights)",
            "vae_decode": {"seconds": round(t_dec, 4), "tflops": round(dec_tf / t_dec / 1e12, 1), "algorithmic_TFLOP": round(dec_tf / 1e12, 1),
                           "frac_of_mfma_peak": round(dec_tf / t_dec / 2.5e15, 3), "what": "latents on the device -> uint8 frames on the device"},
            "clip_vae_encode": {"seconds": round(t_enc, 4)}}


def pmc_traffic(mode: int):
    """HBM bytes per launch of the GEMM kernels of A-gather mode `mode`, from the committed rocprofv3 PMC summary
    (FETCH_SIZE and WRITE_SIZE collected in separate passes by tools/pmc_run.sh; KiB -> bytes, FETCH doubled per the
    gfx950 correction of MI355X_MICROARCH.md): launch-weighted mean over the kernel templates serving that mode.
    PMC counters cannot be collected from inside this process; None when no summary is present."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_1step.csv")))
    if not files:
        return None, None
    tot = n = 0.0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            m = re.match(r"gemm(_huge)?_kernel<(\d+),", row["kernel"])
            if not m or int(m.group(2)) != mode:
                continue
            try:
                b = 2.0 * 1024.0 * float(row["FETCH_SIZE_per_launch"]) + 1024.0 * float(row["WRITE_SIZE_per_launch"])
            except (KeyError, ValueError):
                continue
            k = float(row["launches"])
            tot += b * k
            n += k
    return (round(tot / n) if n else None), os.path.basename(files[-1])


if __name__ == "__main__":
    main()
