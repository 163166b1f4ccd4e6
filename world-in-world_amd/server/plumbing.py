"""Request / response plumbing of the SVD world-model worker (uint8 / int edges of the hot path).

Restates, with citations, the small Python functions of the reference that sit between the wire
protocol and the pipeline:
  * `check_inputdict` / `check_outputdict`      downstream/utils/worker_manager.py:106-147
  * `process_input_dict` / `prepare_image_list` downstream/api_models/__init__.py:185-224
  * `images_to_tensor`                          downstream/api_models/__init__.py:113-166
  * `process_output_dict`                       downstream/api_models/__init__.py:226-237
  * `save_predict` on-disk format               downstream/utils/saver.py:221-275, 415-451
  * `numpy_to_pil` rounding                     dp/image_processor.py:133-147
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence

import numpy as np
from PIL import Image

WORLD_MODEL_NAME = "igen"  # downstream/vlm.py:31; eval_inference.py never defines it
NAV_ACTIONS = {"forward": 1, "turn_left": 2, "turn_right": 3, "stop": 4, "placeholder": 0}


def check_inputdict(d: dict) -> None:
    if not isinstance(d, dict):
        raise AssertionError("request must be a dict")
    missing = [k for k in ("b_action", "save_dirs", "request_model_name") if k not in d]
    if missing:
        raise KeyError(f"Missing required keys: {missing}. Required: ['b_action', 'save_dirs', 'request_model_name']")
    for k, v in d.items():
        if k == "b_image":
            assert isinstance(v, np.ndarray) and v.dtype == np.uint8, "b_image must be a uint8 ndarray"
        elif k == "b_action":
            assert isinstance(v, list) or (isinstance(v, np.ndarray) and v.dtype == np.int64), \
                "b_action must be an int64 ndarray or a list"
        elif k == "save_dirs":
            assert isinstance(v, list) and all(isinstance(s, str) for s in v), "save_dirs should be list[str]"
        elif k == "return_objects":
            assert isinstance(v, list) and all(isinstance(s, bool) for s in v), "return_objects should be list[bool]"


def check_b_action(b_action, num_frames: int, task_type: str = "navigation") -> np.ndarray:
    """Shape facts of `b_action` the reference worker assumes (eval_inference.py:313-331): (b, T) action ids for
    navigation, (b, T, 8) continuous [x, y, z, qx, qy, qz, qw, gripper] rows for manipulation (utils/svd_utils.py:377-409)."""
    b_action = np.asarray(b_action)
    if task_type == "manipulation":
        assert b_action.ndim == 3 and b_action.shape[1:] == (num_frames, 8), \
            f"manipulation b_action must be (b, {num_frames}, 8), got {b_action.shape}"
        # the reference raises inside scipy's Rotation.from_quat on these (utils/svd_utils.py:357-375); here the rotation matrix is
        # closed-form and would carry NaNs into the action embedding: refuse before compute is committed
        a = b_action.astype(np.float64)
        assert np.isfinite(a).all(), "manipulation b_action holds non-finite values"
        assert (np.linalg.norm(a[..., 3:7], axis=-1) > 0).all(), "manipulation b_action holds a zero-norm quaternion"
    else:
        assert b_action.ndim == 2 and b_action.shape[1] == num_frames, \
            f"navigation b_action must be (b, {num_frames}), got {b_action.shape}"
    return b_action


def validate_request(d: dict, num_frames: int, task_type: str = "navigation") -> None:
    """Everything the worker would trip over, checked BEFORE compute is committed (a multi-GPU server must not hand a
    malformed request to its ranks): the reference's `check_inputdict` plus the shape facts its worker assumes
    (`check_b_action`; one save_dir / image per candidate; b_image uint8 (b, C>=3, H, W); `<save_dir>/cond_rgb.png` present
    when no b_image travels)."""
    check_inputdict(d)
    b_action = check_b_action(d["b_action"], num_frames, task_type)
    b = b_action.shape[0]
    assert b > 0 and len(d["save_dirs"]) == b, "one save_dir per candidate"
    if "return_objects" in d:
        assert len(d["return_objects"]) == b, "one return_objects flag per candidate"
    img = d.get("b_image")
    if img is not None:
        assert img.ndim == 4 and img.shape[0] == b and img.shape[1] >= 3, f"b_image should be uint8 (b, 3, H, W), got {img.shape}"
    else:
        missing = [s for s in d["save_dirs"] if not os.path.isfile(os.path.join(s, "cond_rgb.png"))]
        assert not missing, f"no b_image and no cond_rgb.png under {missing[:3]}"


def check_outputdict(d: dict) -> None:
    pf = d.get("pred_frames")
    assert pf is None or (isinstance(pf, np.ndarray) and pf.dtype == np.uint8)
    assert "video_tensors" not in d
    assert isinstance(d["save_dirs"], list)


def parse_request(d: dict, world_model_name: str = WORLD_MODEL_NAME):
    """-> (b_action int64 (b,T), save_dirs, return_objects(bool), images: list of PIL RGB)."""
    # The reference asserts equality (api_models/__init__.py:189), but its manager slices EVERY value of the
    # request with v[start:start+bs] (worker_manager.py:458) — the string too — so behind the manager a worker
    # sees "i", "g", "e", "n", "" ... for candidates 0, 1, 2, ...  Accept exactly those slices of the deployed name.
    name = d["request_model_name"]
    assert isinstance(name, str) and name in world_model_name, (
        f"request_model_name: {name} does not match deployed world_model_name: {world_model_name}")
    b_action = np.asarray(d["b_action"])
    save_dirs = d["save_dirs"]
    assert len(b_action) == len(save_dirs)
    ro = d.get("return_objects")
    return_objects = bool(ro) if ro is not None else False  # truthiness of the list, as in the reference (:204)
    b_image = d.get("b_image")
    if b_image is None:  # the worker loads <save_dir>/cond_rgb.png (:100-110)
        images = [Image.open(os.path.join(s, "cond_rgb.png")).convert("RGB") for s in save_dirs]
    else:
        assert b_image.ndim == 4 and b_image.dtype == np.uint8, f"b_image should be uint8 B C H W, got {b_image.shape}"
        images = [Image.fromarray(np.ascontiguousarray(np.transpose(im[:3], (1, 2, 0)))) for im in b_image]
    return b_action, save_dirs, return_objects, images


def preprocess_image(img: Image.Image, width: int, height: int) -> np.ndarray:
    """VideoProcessor.preprocess (pipeline:521): PIL LANCZOS resize to (width, height), [0,1] -> [-1,1];
    returns float32 (3, H, W)."""
    if img.size != (width, height):
        img = img.resize((width, height), Image.LANCZOS)
    x = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return np.transpose(2.0 * x - 1.0, (2, 0, 1))


def image_to_array(img: Image.Image) -> np.ndarray:
    """pil_to_numpy + numpy_to_pt + `* 2 - 1` of `_encode_image` (pipeline:192-199): float32 (3, H0, W0) in [-1,1] at the
    image's OWN size (no resize)."""
    x = np.asarray(img.convert("RGB"), dtype=np.float32) / 255.0
    return np.transpose(2.0 * x - 1.0, (2, 0, 1))


def frames_to_pil(frames: np.ndarray) -> List[Image.Image]:
    """decoded frames (T,3,H,W) in [-1,1] -> PIL list: denormalise, clamp, (x*255).round() (image_processor.py:147)."""
    x = np.clip(frames / 2.0 + 0.5, 0.0, 1.0)
    u8 = (np.transpose(x, (0, 2, 3, 1)) * 255).round().astype("uint8")
    return [Image.fromarray(f) for f in u8]


def images_to_tensor(pipe_images: Sequence[Sequence[Image.Image]], save_size=(480, 480)) -> np.ndarray:
    """PIL BICUBIC resize to save_size=(W,H) then /255, CHW  ->  float32 (B,T,3,H,W) (api_models/__init__.py:113-166)."""
    def one(im):
        return np.transpose(np.asarray(im.resize(save_size, Image.BICUBIC), dtype=np.float32) / 255.0, (2, 0, 1))

    flat = [im for clip in pipe_images for im in clip]
    # PIL's resize releases the GIL: the frames of a request are resized on a small thread pool (same result per
    # frame; 14 x B frames of 576x1024 cost 0.16 s x B on one thread, next to 0.17 s x B of VAE decode on the GPU)
    fr = list(_pool().map(one, flat)) if len(flat) > 1 else [one(im) for im in flat]
    it = iter(fr)
    return np.stack([np.stack([next(it) for _ in clip]) for clip in pipe_images])


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1)))
    return _POOL


def save_predict(video: np.ndarray, b_action, save_dirs: Sequence[str]) -> None:
    """<dir>/<i>.jpg (torchvision save_image rounding: x*255+0.5 clamp, JPEG default quality) + action_seq.json."""
    for clip, d, act in zip(video, save_dirs, b_action):
        os.makedirs(d, exist_ok=True)
        for i, fr in enumerate(clip):
            u8 = np.clip(np.transpose(fr, (1, 2, 0)) * 255.0 + 0.5, 0, 255).astype(np.uint8)
            path = os.path.join(d, f"{i}.jpg")
            Image.fromarray(u8).save(path)
            try:
                os.chmod(path, 0o666)
            except OSError:
                pass
        with open(os.path.join(d, "action_seq.json"), "w") as f:
            json.dump(np.asarray(act).tolist(), f, indent=2, ensure_ascii=False)


def build_response(video: np.ndarray, b_action, save_dirs: List[str], return_objects: bool) -> dict:
    """process_output_dict (api_models/__init__.py:226-237): uint8 by TRUNCATION of clip(x,0,1)*255."""
    video = np.asarray(video, dtype=np.float32)
    if return_objects:
        return {"pred_frames": (np.clip(video, 0, 1) * 255).astype(np.uint8), "save_dirs": save_dirs}
    save_predict(video, b_action, save_dirs)
    return {"save_dirs": save_dirs}


def split_batch(tasks: dict, bs: int):
    """Batcher.split_batch (worker_manager.py:448-469): every value is sliced v[start:start+bs]."""
    n = len(next(iter(tasks.values())))
    assert n > 0
    return [{k: v[s:s + bs] for k, v in tasks.items()} for s in range(0, n, bs)]


def recompose(results: Sequence[dict]) -> dict:
    """Batcher._recompose_batch (:471-481): list.extend per key in ascending task id."""
    out: dict = {}
    for item in results:
        for k, v in item.items():
            out.setdefault(k, []).extend(v)
    return out
