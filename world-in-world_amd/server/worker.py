"""The SVD world-model worker: request dict -> 14-frame rollouts -> response dict.

Mirrors `do_some_tasks` / `Navigator.inference` of the reference worker
(FTsvd/eval_inference.py:228-266, 313-349) with the same fixed knobs (fps 7, motion bucket 127,
noise_aug 0.02, 30 Euler steps served / 25 in the benchmark metric, output 480x480) and the same
persistent seeded generator semantics (one generator for the worker's lifetime, :97, 258; draw order
image-noise then latent-noise).

The denoising loop runs on the HIP kernels (`SVDDenoiser`).  Image conditioning (CLIP embed + VAE
encode) and latent decoding (temporal VAE) come through `frontend` — in the product `vae.HIPFrontend`,
whose VAE encode / temporal decode run on the same HIP kernels; it is injected so that the worker is
testable without checkpoints (and with the fp32 PyTorch checker of oracle/vae_oracle.py).

Two transports:
  * `serve_tcp`   — speaks the client protocol directly (what `Solver.send_batch_to_server` expects,
                    downstream/solver_base.py:645-688): one request -> one response, "DONE" closes;
  * `worker_main` — manager-compatible loop: framed `(client_id, task_id, payload)` on stdin,
                    framed `(client_id, task_id, result)` on the inherited fd given as last argv
                    (downstream/utils/worker_manager.py:660-702).
"""
from __future__ import annotations

import argparse
import os
import socket
import sys
import threading
from typing import Callable, Optional, Protocol

import numpy as np

from . import plumbing as P
from .protocol import DONE, read_framed, read_pickled, write_framed, write_pickled


class Frontend(Protocol):
    """PyTorch side of the worker (CLIP + VAE)."""

    def encode(self, images: np.ndarray, image_noise: np.ndarray, noise_aug_strength: float, clip_images=None):
        """images (B,3,H,W) in [-1,1] (LANCZOS-resized to the model size: the VAE branch, pipeline:521);
        clip_images: optional list of B arrays (3,H0,W0) in [-1,1] at the ORIGINAL size — what the reference's
        `_encode_image` sees (pipeline:192-199 runs on the un-resized PIL image); None = same as `images`.
        Returns (image_latents (B,4,h,w), image_embeddings (B,1,D)) float32."""

    def decode(self, latents: np.ndarray) -> np.ndarray:
        """latents (B,T,4,h,w) -> frames (B,T,3,H,W) in [-1,1] float32 (decode_latents, pipeline:282-309)."""


class SVDWorker:
    def __init__(self, denoise_fn: Callable[..., "np.ndarray"], frontend: Frontend, *, width=1024, height=576,
                 out_width=480, out_height=480, num_frames=14, num_inference_steps=30, seed=1,
                 world_model_name=P.WORLD_MODEL_NAME, noise_fn: Optional[Callable] = None, task_type: str = "navigation"):
        self.denoise_fn = denoise_fn
        self.frontend = frontend
        self.width, self.height = width, height
        self.out_size = (out_width, out_height)
        self.num_frames = num_frames
        self.task_type = task_type      # 'navigation': (b, T) action ids; 'manipulation': (b, T, 8) continuous actions
        self.num_inference_steps = num_inference_steps
        self.world_model_name = world_model_name
        # persistent generator: the draws of request k depend on requests 0..k-1, as in the reference
        self._rng = np.random.Generator(np.random.Philox(seed))
        self.noise_fn = noise_fn or (lambda shape: self._rng.standard_normal(shape, dtype=np.float32))
        self.bind_thread: Optional[Callable[[], None]] = None   # set by the launcher: per-thread HIP device binding

    def __call__(self, request: dict) -> dict:
        """do_some_tasks (eval_inference.py:313-349)."""
        if self.bind_thread is not None:
            self.bind_thread()
        b_action, save_dirs, return_objects, images = P.parse_request(request, self.world_model_name)
        b_action = P.check_b_action(b_action, self.num_frames, self.task_type)
        B = len(images)
        x = np.stack([P.preprocess_image(im, self.width, self.height) for im in images])
        # CLIP sees the image at its ORIGINAL size (pipeline:192-199); only the VAE branch is resized (pipeline:521)
        clip_images = None
        if any(im.size != (self.width, self.height) for im in images):
            clip_images = [P.image_to_array(im) for im in images]
        img_noise = self.noise_fn(x.shape)                                   # draw 1 (pipeline:522)
        if clip_images is None:
            image_latents, image_embeddings = self.frontend.encode(x, img_noise, 0.02)
        else:
            image_latents, image_embeddings = self.frontend.encode(x, img_noise, 0.02, clip_images=clip_images)
        h, w = image_latents.shape[-2:]
        lat_noise = self.noise_fn((B, self.num_frames, 4, h, w))             # draw 2 (pipeline:765)
        latents = self.denoise_fn(image_latents, image_embeddings, lat_noise, b_action,
                                  num_steps=self.num_inference_steps, fps=7, motion_bucket_id=127,
                                  noise_aug_strength=0.02)
        if hasattr(self.frontend, "decode_uint8"):    # frame quantisation on the device (same arithmetic, same bytes);
            # `latents` may be a device tensor: nothing crosses PCIe between the loop and the decoder
            clips = [[P.Image.fromarray(f) for f in clip] for clip in self.frontend.decode_uint8(latents)]
        else:
            lat = np.asarray(latents.cpu() if hasattr(latents, "cpu") else latents, dtype=np.float32)
            frames = self.frontend.decode(lat)         # (B,T,3,H,W) in [-1,1]
            clips = [P.frames_to_pil(f) for f in frames]
        video = P.images_to_tensor(clips, save_size=self.out_size)
        out = P.build_response(video, b_action, list(save_dirs), return_objects)
        P.check_outputdict(out)
        return out


# ------------------------------------------------------------------------------------------------
# cross-client batching (the manager's batching policy, done so that B = 8 per GPU is FORMED)
# ------------------------------------------------------------------------------------------------
class Coalescer:
    """Forms GPU batches ACROSS clients.  The reference manager does the opposite: it splits every request into
    batch-1 tasks (worker_manager.py:448-469), sleeps 60 ms per dispatched task (:570) and polls results every 50 ms
    (:548), so a worker never sees more than one candidate.  Here every client handler `submit`s its request; one
    compute thread takes the oldest pending request, keeps collecting for at most `max_wait_s` or until `max_candidates`
    candidates are pending, concatenates the requests key by key (the same `v[lo:hi]` algebra the manager uses to split,
    inverted), runs the worker ONCE, and hands every client exactly its own candidates back, in order.  Wire format and
    per-candidate results are unchanged (candidates are evaluated independently of what else is in the batch — the B >= 2
    contract, bit for bit on the HIP path).

    Failure containment: if the merged call raises, the participating requests are re-run one by one, so only the
    offending client sees the error."""

    def __init__(self, worker: Callable[[dict], dict], max_candidates: int = 8, max_wait_s: float = 0.02):
        self.worker = worker
        self.max_candidates = max(1, int(max_candidates))
        self.max_wait_s = float(max_wait_s)
        self._cv = threading.Condition()
        self._pending: list = []          # [request, n_candidates, result slot (dict), done Event]
        self._stop = False
        self.batches: list = []           # candidates per worker call (observability / tests)
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    @staticmethod
    def _count(req: dict) -> int:
        return len(req["save_dirs"])

    @staticmethod
    def merge(reqs: list) -> dict:
        out: dict = {}
        keys = set(reqs[0])
        for r in reqs[1:]:
            if set(r) != keys:
                raise ValueError("requests with different key sets cannot share a batch")
        for k in reqs[0]:
            vs = [r[k] for r in reqs]
            if isinstance(vs[0], str):
                if any(v != vs[0] for v in vs):
                    raise ValueError(f"{k} differs between requests")
                out[k] = vs[0]
            elif isinstance(vs[0], np.ndarray):
                out[k] = np.concatenate([np.asarray(v) for v in vs])
            else:
                out[k] = [x for v in vs for x in v]
        return out

    @staticmethod
    def split(resp: dict, counts: list) -> list:
        outs, lo = [], 0
        for n in counts:
            outs.append({k: (v if isinstance(v, str) else v[lo:lo + n]) for k, v in resp.items()})
            lo += n
        return outs

    def submit(self, req: dict) -> dict:
        slot: dict = {}
        done = threading.Event()
        with self._cv:
            self._pending.append([req, self._count(req), slot, done])
            self._cv.notify_all()
        done.wait()
        if "error" in slot:
            raise slot["error"]
        return slot["resp"]

    def close(self) -> None:
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._thread.join(5)

    def _take_batch(self):
        import time
        with self._cv:
            while not self._pending and not self._stop:
                self._cv.wait(0.2)
            if self._stop and not self._pending:
                return None
            deadline = time.monotonic() + self.max_wait_s
            while sum(p[1] for p in self._pending) < self.max_candidates:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                self._cv.wait(left)
            batch, total = [], 0
            while self._pending and (not batch or total + self._pending[0][1] <= self.max_candidates):
                item = self._pending.pop(0)
                batch.append(item)
                total += item[1]
            return batch

    def _run(self):
        while True:
            batch = self._take_batch()
            if batch is None:
                return
            try:
                merged = self.merge([b[0] for b in batch]) if len(batch) > 1 else batch[0][0]
                self.batches.append(sum(b[1] for b in batch))
                resp = self.worker(merged)
                parts = self.split(resp, [b[1] for b in batch]) if len(batch) > 1 else [resp]
                for b, part in zip(batch, parts):
                    b[2]["resp"] = part
            except BaseException:  # noqa: BLE001 — isolate the offender: one request at a time
                for b in batch:
                    if len(batch) == 1:
                        import sys as _s
                        b[2]["error"] = _s.exc_info()[1]
                        continue
                    try:
                        self.batches.append(b[1])
                        b[2]["resp"] = self.worker(b[0])
                    except BaseException as e:  # noqa: BLE001
                        b[2]["error"] = e
            for b in batch:
                b[3].set()


# ------------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------------
def _handle_client(conn: socket.socket, worker: Callable[[dict], dict], batch_size: int, lock: threading.Lock,
                   coalescer: Optional[Coalescer] = None):
    with conn:
        while True:
            try:
                req = read_framed(conn)
            except EOFError:
                return
            if isinstance(req, str) and req == DONE:
                return
            P.check_inputdict(req)
            if coalescer is not None:
                write_framed(conn, coalescer.submit(req))
                continue
            if batch_size and batch_size > 0:  # the manager's split / recompose (worker_manager.py:448-481)
                parts = []
                for sub in P.split_batch(req, batch_size):
                    with lock:
                        parts.append(worker(sub))
                resp = P.recompose(parts)
            else:
                with lock:
                    resp = worker(req)
            write_framed(conn, resp)


def serve_tcp(worker: Callable[[dict], dict], host="127.0.0.1", port=7000, batch_size: int = 0,
              ready: Optional[threading.Event] = None, stop: Optional[threading.Event] = None,
              coalesce_candidates: int = 0, coalesce_wait_s: float = 0.02) -> None:
    """Accept loop (worker_manager.py:644-656): one thread per client, compute serialised by a lock.
    batch_size > 0 reproduces the manager's split into sub-batches (responses then carry LISTS per key,
    exactly what the reference's recompose yields); 0 hands the whole request to the worker (true batching).
    coalesce_candidates > 0: requests of DIFFERENT clients that arrive within `coalesce_wait_s` are evaluated as one
    batch of up to that many candidates (`Coalescer`)."""
    lock = threading.Lock()
    coalescer = Coalescer(worker, coalesce_candidates, coalesce_wait_s) if coalesce_candidates > 0 else None
    if ready is not None:
        ready.coalescer = coalescer
    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind((host, port))
    srv.listen(16)
    srv.settimeout(0.2)
    if ready is not None:
        ready.port = srv.getsockname()[1]
        ready.set()
    try:
        while stop is None or not stop.is_set():
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                continue
            threading.Thread(target=_handle_client, args=(conn, worker, batch_size, lock, coalescer), daemon=True).start()
    finally:
        srv.close()
        if coalescer is not None:
            coalescer.close()


def worker_main(pipe_fd: int, task_fn: Callable[[dict], dict], stdin=None) -> None:
    """Manager-compatible worker loop (worker_manager.py:660-702): strictly serial, exceptions are NOT
    swallowed (the reference lets the process die; the manager then logs EOF)."""
    stdin = stdin or sys.stdin.buffer
    with os.fdopen(pipe_fd, "wb", buffering=2048 * 1024) as out:
        while True:
            try:
                item = read_pickled(stdin)
            except EOFError:
                break
            if isinstance(item, str) and item == DONE:
                break
            client_id, task_id, payload = item
            if isinstance(payload, str) and payload == DONE:
                break
            result = task_fn(payload)
            P.check_outputdict(result)
            write_pickled(out, (client_id, task_id, result))


def build_arg_parser() -> argparse.ArgumentParser:
    """CLI surface the reference launcher passes (eval_inference.py:273-295, workers_cfg.py:23-30, 270-280)."""
    ap = argparse.ArgumentParser(description="MI355X-native SVD world-model worker")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--out_width", type=int, default=480)
    ap.add_argument("--out_height", type=int, default=480)
    ap.add_argument("--log_dir", type=str, default="downstream/logs")
    ap.add_argument("--exp_id", type=str, default="wiw_amd")
    ap.add_argument("--num_frames", type=int, default=14)
    ap.add_argument("--num_past_obs", type=int, default=1)
    ap.add_argument("--task_type", type=str, default="navigation", choices=["navigation", "manipulation"],
                    help="manipulation: b_action rows are [x, y, z, qx, qy, qz, qw, gripper] (b, 14, 8); pass "
                         "--action_input_channel 10 (or 23, the positional form) as the checkpoint was trained")
    ap.add_argument("--action_strategy", type=str, default="micro_cond", choices=["micro_cond"])
    ap.add_argument("--action_input_channel", type=int, default=14)
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--unet_path", type=str, default="")
    ap.add_argument("--svd_path", type=str, default="")
    ap.add_argument("--weight_dtype", type=str, default="float16",
                    help="16-bit storage / MFMA operand type: float16 (default: the reference worker's own, FTsvd/eval_inference.py:294) "
                         "or bfloat16 (the dtype BASELINE.json quotes its metric in; what bench.py measures by default)")
    ap.add_argument("--num_inference_steps", type=int, default=30)
    ap.add_argument("--port", type=int, default=0, help="> 0: standalone TCP server instead of the manager pipe loop")
    ap.add_argument("--hip_graph", action=argparse.BooleanOptionalAction, default=True,
                    help="replay the UNet forward from a captured hipGraph (one per candidate count; same bytes, ~12 ms less host "
                         "work per forward; a captured shape pins ~6 GB of HBM per candidate at 576x1024).  ON by default — the "
                         "mode bench.py measures; --no-hip_graph launches eagerly")
    ap.add_argument("--residual_fp32", action=argparse.BooleanOptionalAction, default=None,
                    help="keep the UNet's block-level residual stream in fp32 (UNetHIP(residual_fp32=True); the reference keeps latents "
                         "and the Euler step in fp32, scheduling_euler_discrete.py:635,673).  Default: ON with --weight_dtype float16 — "
                         "the configuration whose 25-step latents at 576x1024x14 are within 1e-3 of the reference pipeline's in BOTH "
                         "norms (5.5e-4 rms / 8.6e-4 max; plain fp16: 6.7e-4 / 1.09e-3; tests/test_hip_res32.py, DESIGN.md 5) at 0.96x "
                         "the speed of plain fp16 — and OFF with bfloat16 (whose own rounding floor is 5e-3).  --no-residual_fp32: off")
    ap.add_argument("--residual_fp32_full", action="store_true",
                    help="with --residual_fp32: also the hidden stream inside the transformer blocks in fp32 (rounds 4-5's form of "
                         "the mode: 5.1e-4 / 8.6e-4 on the same trajectory at 0.90x the speed of fp16)")
    ap.add_argument("--batch_size", type=int, default=0)
    ap.add_argument("--coalesce_candidates", type=int, default=0,
                    help="> 0: batch requests of different clients into one GPU call of up to this many candidates")
    ap.add_argument("--coalesce_wait_ms", type=float, default=20.0, help="how long the oldest pending request may wait for company")
    return ap


def resolve_precision(args):
    """-> (dtype name, residual_fp32 of UNetHIP) from --weight_dtype / --residual_fp32 / --residual_fp32_full.  The drop-in
    default (float16, the reference worker's dtype, eval_inference.py:294) carries the block-level fp32 residual stream unless
    told otherwise: the configuration whose 25-step latents at 576x1024x14 are within 1e-3 of the reference pipeline's in both
    norms; bfloat16 (BASELINE's dtype; rounding floor 5e-3) does not.  fp32 is not a serving dtype of this path."""
    names = {"bfloat16": "bfloat16", "bf16": "bfloat16", "torch.bfloat16": "bfloat16",
             "float16": "float16", "fp16": "float16", "half": "float16", "torch.float16": "float16"}
    if args.weight_dtype not in names:
        raise SystemExit(f"--weight_dtype {args.weight_dtype!r}: the HIP path serves bfloat16 or float16")
    dtype = names[args.weight_dtype]
    res32 = (dtype == "float16") if args.residual_fp32 is None else bool(args.residual_fp32)
    if getattr(args, "residual_fp32_full", False):
        if not res32:
            raise SystemExit("--residual_fp32_full extends --residual_fp32 (on by default with float16 only)")
        res32 = "full"
    return dtype, res32


def validate_args(args) -> None:
    """Configurations this build does not serve are refused at START-UP, not at the first client request (on every rank of
    a sharded server): --num_past_obs > 1 — the WM-server request has no field for past observations and the reference's
    own served path asserts there are none (eval_inference.py:236-242, 343-348: `base_img_path=pil_images` -> `past_obs_pixel is
    None`); the general Sk > 1 cross-attention exists below the server (UNetHIP(num_past_obs=P), csrc/cross_attn.hip) for the
    offline / training callers that build (B, P, 1024) embeddings themselves (train_svd.py:889-894); an action
    embedder width the task cannot produce (navigation: one channel per frame, get_action_ids micro_cond; manipulation: 10 =
    [norm_xyz | r6 | norm_grip] or num_frames + 9, its positional form — utils/svd_utils.py:418-457, 499-567)."""
    if args.num_past_obs != 1:
        raise SystemExit(f"--num_past_obs {args.num_past_obs}: the WM-server protocol carries ONE conditioning image per candidate "
                         f"(the reference's served path asserts past_obs_pixel is None, eval_inference.py:236-242); P > 1 tokens are "
                         f"served by UNetHIP(num_past_obs=P) below the server only")
    if args.task_type == "manipulation":
        ok = (10, args.num_frames + 9)
        if args.action_input_channel not in ok:
            raise SystemExit(f"--task_type manipulation needs --action_input_channel {ok[0]} or {ok[1]} (num_frames + 9), "
                             f"got {args.action_input_channel}")
    elif args.action_input_channel != args.num_frames:
        raise SystemExit(f"--task_type navigation embeds one action channel per frame: --action_input_channel must equal "
                         f"--num_frames ({args.num_frames}), got {args.action_input_channel}")
