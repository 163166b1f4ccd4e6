#!/usr/bin/env python
"""Launcher for the MI355X SVD world-model worker: the process the reference manager starts in place of
FTsvd/eval_inference.py (downstream/utils/worker_manager.py:324-334 spawns `<python> <script> <args...> <w_fd>`),
or a standalone TCP server with --port (client protocol of downstream/solver_base.py:645-688).

    python serve_worker.py --unet_path <finetuned checkpoint dir (holds unet/)> --svd_path <stable-video-diffusion-img2vid-xt> <w_fd>
    python serve_worker.py --unet_path ... --svd_path ... --port 7000
    python serve_worker.py --random_weights --port 7000          # no checkpoints: random-init weights (bring-up)
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 serve_worker.py --port 7000 ...   # one request over 8 GPUs

Everything between request ingest and the uint8 frames runs on the hand-written HIP path (libwiwsvd.so; no fallback): the
CLIP ViT-H/14 image encoder (`clip.CLIPVisionHIP` — the `transformers` module is only the container its weights are read
from), the VAE encoder, the denoising loop and the temporal VAE decoder.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

import wiw_amd  # noqa: F401  (registers the package under an importable name)
from wiw_amd import frontend as FE
from wiw_amd.config import UNetConfig
from wiw_amd.pipeline import SVDDenoiser
from wiw_amd.server.worker import SVDWorker, build_arg_parser, resolve_precision, serve_tcp, validate_args, worker_main
from wiw_amd.unet import UNetHIP
from wiw_amd.vae import HIPFrontend, VAEHIP
from wiw_amd.weights import load_safetensors, random_state_dict


WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors")


def _find_safetensors(*folders: str) -> str:
    """First `diffusion_pytorch_model[.fp16].safetensors` found in `folders` (searched in order)."""
    for folder in folders:
        for name in WEIGHT_FILES:
            p = os.path.join(folder, name)
            if os.path.isfile(p):
                return p
    raise FileNotFoundError(f"no diffusion_pytorch_model[.fp16].safetensors under any of {list(folders)}")


def resolve_unet_weights(unet_path: str, svd_path: str) -> str:
    """Where the reference finds the UNet: `from_pretrained(unet_path, subfolder='unet')` (eval_inference.py:115-131) —
    the manager passes `--unet_path=.../checkpoint-6000` (workers_cfg.py:26), so the file is `<unet_path>/unet/...`;
    a path that already IS the unet folder is accepted too.  A non-empty --unet_path that does not exist is an ERROR
    (no silent fallback to the stock model); an empty one means the stock `<svd_path>/unet`."""
    if unet_path:
        if not os.path.isdir(unet_path):
            raise FileNotFoundError(f"--unet_path {unet_path!r} is not a directory")
        return _find_safetensors(os.path.join(unet_path, "unet"), unet_path)
    return _find_safetensors(os.path.join(svd_path, "unet"))


def _clip(svd_path: str, random_weights: bool):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    if random_weights:   # ViT-H/14 geometry of the SVD image_encoder
        torch.manual_seed(0)
        return CLIPVisionModelWithProjection(CLIPVisionConfig(
            hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
            patch_size=14, projection_dim=1024, hidden_act="gelu")).eval()
    enc = os.path.join(svd_path, "image_encoder")
    # the reference pipeline loads with variant='fp16' and local_files_only (eval_inference.py:134-141): an fp16-only
    # snapshot carries model.fp16.safetensors and no plain file
    variant = "fp16" if (not os.path.isfile(os.path.join(enc, "model.safetensors")) and
                         os.path.isfile(os.path.join(enc, "model.fp16.safetensors"))) else None
    return CLIPVisionModelWithProjection.from_pretrained(svd_path, subfolder="image_encoder", variant=variant,
                                                         local_files_only=True).eval()


def arg_parser():
    ap = build_arg_parser()
    ap.add_argument("--random_weights", action="store_true", help="random-init UNet / VAE / CLIP (no checkpoints)")
    ap.add_argument("--host", type=str, default="127.0.0.1")
    ap.add_argument("pipe_fd", nargs="?", type=int, default=None, help="result-pipe fd appended by the manager")
    return ap


def build_worker(args, cfg: UNetConfig = None, vae_cfg: dict = None, clip=None) -> SVDWorker:
    """cfg / vae_cfg / clip: overrides for tests (reduced-width checkpoints); the served geometry by default."""
    if cfg is None:
        validate_args(args)
    cfg = cfg or UNetConfig(num_frames=args.num_frames, action_input_channel=args.action_input_channel, task_type=args.task_type)
    vae_cfg = vae_cfg or {}
    if args.random_weights:
        unet_sd, vae_sd = random_state_dict(cfg, 0), FE.vae_random_state_dict(1, **vae_cfg)
    else:
        unet_sd = load_safetensors(resolve_unet_weights(args.unet_path, args.svd_path))
        vae_sd = load_safetensors(_find_safetensors(os.path.join(args.svd_path, "vae")))
    torch.cuda.set_device(torch.device(args.device))   # the C ABI sizes its grids from the CURRENT device
    # 16-bit storage / MFMA operand type of UNet, VAE and CLIP: fp16 (the reference's own default, eval_inference.py:294, and
    # the default here since round 6) or bf16 (BASELINE's dtype) — libwiwsvd_f16.so / libwiwsvd.so.  fp32 is not a serving dtype of
    # this path (the reference upcasts only the VAE encoder, pipeline:525-527).
    dname, res32 = resolve_precision(args)     # float16 + the block-level fp32 residual stream unless told otherwise
    dtype = getattr(torch, dname)
    unet = UNetHIP(cfg, unet_sd, args.device, dtype=dtype, residual_fp32=res32)
    den = SVDDenoiser(unet, use_graph=bool(args.hip_graph))   # graph replay by default, as bench.py measures
    clip = clip if clip is not None else _clip(args.svd_path, args.random_weights)
    # VAE and CLIP on the HIP kernels (vae.py, clip.py: HIPFrontend builds `CLIPVisionHIP` from the `transformers` module's
    # weights).  There is no PyTorch / MIOpen route in the product (the VAE alone needed > 6 minutes per decode on a fresh
    # box); the fp32 PyTorch chain lives in oracle/.
    fe = HIPFrontend(VAEHIP(vae_sd, args.device, hip=unet.hip, **vae_cfg), clip, dtype=dtype, device_io=True)

    def denoise(image_latents, image_embeddings, noise, actions, **kw):
        # device tensors in (HIPFrontend.encode, device_io) and out (decode_uint8 takes them): no host bounce of latents
        return den.denoise(torch.as_tensor(image_latents), torch.as_tensor(image_embeddings), torch.as_tensor(noise),
                           actions, **kw)

    worker = SVDWorker(denoise, fe, width=args.width, height=args.height, out_width=args.out_width,
                       out_height=args.out_height, num_frames=args.num_frames,
                       num_inference_steps=args.num_inference_steps, task_type=cfg.task_type)
    worker.bind_thread = unet.hip.bind_thread     # serve_tcp handler threads start on HIP device 0
    return worker


def main(argv=None) -> None:
    ap = arg_parser()
    args = ap.parse_args(list(sys.argv[1:] if argv is None else argv))
    if args.port <= 0 and args.pipe_fd is None:
        ap.error("either --port (standalone TCP server) or the manager's trailing pipe fd is required")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:   # torchrun --nproc-per-node N serve_worker.py --port P ...: one request sharded over N GPUs
        import torch.distributed as dist
        from wiw_amd.parallel import ShardedWorker
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        args.device = f"cuda:{local_rank}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(args.device))
        from wiw_amd.server import plumbing as P
        sharded = ShardedWorker(build_worker(args), validate=lambda r: P.validate_request(r, args.num_frames, args.task_type))
        if dist.get_rank() == 0:
            if args.port <= 0:
                ap.error("multi-GPU serving needs --port (rank 0 runs the TCP server)")
            print(f"[serve_worker] {world} ranks, listening on {args.host}:{args.port}", file=sys.stderr, flush=True)
            try:
                serve_tcp(sharded, host=args.host, port=args.port, batch_size=0,
                          coalesce_candidates=args.coalesce_candidates, coalesce_wait_s=args.coalesce_wait_ms / 1e3)
            finally:
                sharded.close()
        else:
            sharded.follow()
        dist.destroy_process_group()
        return
    worker = build_worker(args)
    if args.port > 0:
        print(f"[serve_worker] listening on {args.host}:{args.port}", file=sys.stderr, flush=True)
        serve_tcp(worker, host=args.host, port=args.port, batch_size=args.batch_size,
                  coalesce_candidates=args.coalesce_candidates, coalesce_wait_s=args.coalesce_wait_ms / 1e3)
    else:
        worker_main(args.pipe_fd, worker)


if __name__ == "__main__":
    main()
