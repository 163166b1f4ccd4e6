#!/usr/bin/env python
"""Pin the drop-in boundary to bytes PRODUCED BY THE REFERENCE (build container only).

TEST INFRASTRUCTURE.  Imports /root/reference/downstream/utils/worker_manager.py read-only (its imports are
stdlib + numpy + torch) and records what ITS functions write / decide:

  wire_request.bin        the bytes its `write_framed` puts on a TCP socket for a client request (captured on a socketpair)
  wire_done.bin           the same for the session terminator "DONE"
  wire_task.bin           the bytes its `write_pickled_data` puts on a worker's stdin pipe for (client_id, task_id, payload)
  wire_result.bin         the bytes its `write_pickled_data` would carry back for THIS REPO's response dict
                          (`wiw_amd.server.plumbing.build_response`), after its `check_outputdict` accepted that dict
  wire_batcher.bin        pickle of {bs: (sub-batches of its `Batcher.split_batch`, output of its `_recompose_batch` fed with
                          this repo's per-sub-batch responses)} for bs = 1, 2, 3
  wire_verdicts.bin       pickle of [(case name, dict, verdict of its check_inputdict / check_outputdict)] — verdict is "ok" or
                          the exception class name — incl. this repo's request / response dicts and malformed ones

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_wire_golden.py

tests/test_wire_golden.py replays them through wiw_amd/server/{protocol,plumbing}.py.  The .bin files are pickles of
plain Python / numpy objects (data, not code).
"""
import contextlib
import importlib.util
import io
import os
import pickle
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

import numpy as np  # noqa: E402

import wiw_amd  # noqa: E402,F401
from wiw_amd.server import plumbing as P  # noqa: E402

REF = os.environ.get("WIW_REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_manager():
    path = os.path.join(REF, "downstream", "utils", "worker_manager.py")
    spec = importlib.util.spec_from_file_location("ref_worker_manager", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def capture_framed(wm, obj) -> bytes:
    a, b = socket.socketpair()
    try:
        wm.write_framed(a, obj)
        a.shutdown(socket.SHUT_WR)
        buf = bytearray()
        while True:
            c = b.recv(1 << 20)
            if not c:
                break
            buf += c
        return bytes(buf)
    finally:
        a.close()
        b.close()


def capture_pickled(wm, obj) -> bytes:
    f = io.BytesIO()
    wm.write_pickled_data(f, obj)
    return f.getvalue()


def request(b=3, h=16, w=32, seed=0):
    rs = np.random.RandomState(seed)
    acts = np.array([[4, 1, 2, 1, 3, 1, 1, 2, 2, 1, 3, 3, 1, 1], [4, 3, 3, 3, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1],
                     [4, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]], dtype=np.int64)[:b]
    return {"b_action": acts, "save_dirs": [f"/tmp/wiw_wire/cand_{i}" for i in range(b)], "request_model_name": "igen",
            "b_image": rs.randint(0, 256, size=(b, 3, h, w), dtype=np.uint8), "return_objects": [True] * b}


def our_response(sub: dict, seed: int) -> dict:
    """This repo's response for a sub-request: plumbing.build_response on a synthetic (b,14,3,8,8) video in [0,1]."""
    b = len(sub["save_dirs"])
    video = np.random.RandomState(100 + seed).uniform(-0.1, 1.1, size=(b, 14, 3, 8, 8)).astype(np.float32)
    return P.build_response(video, sub["b_action"], list(sub["save_dirs"]), True)


def verdict(fn, d):
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            fn(d)
        return "ok"
    except Exception as e:  # the class is what a caller can react to
        return type(e).__name__


def main():
    wm = load_reference_manager()
    os.makedirs(OUT, exist_ok=True)
    req = request()

    def dump(name, data: bytes):
        with open(os.path.join(OUT, name), "wb") as f:
            f.write(data)
        print(f"wrote tests/golden/{name} ({len(data)} bytes)")

    dump("wire_request.bin", capture_framed(wm, req))
    dump("wire_done.bin", capture_framed(wm, "DONE"))
    task = (7, 42, {k: v[1:2] for k, v in req.items()})      # what its ClientHandler puts on a worker's stdin (:567)
    dump("wire_task.bin", capture_pickled(wm, task))
    resp = our_response(task[2], 0)
    wm.check_outputdict(resp)                                  # the reference ACCEPTS this repo's response
    dump("wire_result.bin", capture_pickled(wm, (7, 42, resp)))

    batcher = {}
    for bs in (1, 2, 3):
        with contextlib.redirect_stdout(io.StringIO()):
            bt = wm.Batcher(bs, "world_model", ("127.0.0.1", 0), 0)
            subs, next_id = bt.split_batch(req, 0)
            results = [(tid, our_response(sub, tid)) for tid, sub in subs]
            for tid, r in reversed(results):                   # arrival order != task order: it sorts by task id (:474)
                bt.put_new_result(tid, r)
            out = bt.get()
        batcher[bs] = dict(subs=subs, next_id=next_id, responses=results, recomposed=out)
    dump("wire_batcher.bin", pickle.dumps(batcher, protocol=4))

    cases = [("request", req, "in"), ("request_no_image", {k: v for k, v in req.items() if k != "b_image"}, "in"),
             ("request_list_actions", dict(req, b_action=req["b_action"].tolist()), "in"),
             ("request_missing_key", {k: v for k, v in req.items() if k != "save_dirs"}, "in"),
             ("request_float_actions", dict(req, b_action=req["b_action"].astype(np.float32)), "in"),
             ("request_int32_actions", dict(req, b_action=req["b_action"].astype(np.int32)), "in"),
             ("request_float_image", dict(req, b_image=req["b_image"].astype(np.float32)), "in"),
             ("request_save_dirs_tuple", dict(req, save_dirs=tuple(req["save_dirs"])), "in"),
             ("request_return_objects_ints", dict(req, return_objects=[1, 1, 1]), "in"),
             ("request_not_a_dict", [1, 2, 3], "in"),
             ("response", resp, "out"), ("response_files_only", {"save_dirs": resp["save_dirs"]}, "out"),
             ("response_float_frames", dict(resp, pred_frames=resp["pred_frames"].astype(np.float32)), "out"),
             ("response_list_frames", dict(resp, pred_frames=list(resp["pred_frames"])), "out"),
             ("response_video_tensors", dict(resp, video_tensors=np.zeros(1)), "out"),
             ("response_save_dirs_tuple", dict(resp, save_dirs=tuple(resp["save_dirs"])), "out"),
             ("response_no_save_dirs", {"pred_frames": resp["pred_frames"]}, "out")]
    verdicts = [(name, d, kind, verdict(wm.check_inputdict if kind == "in" else wm.check_outputdict, d)) for name, d, kind in cases]
    for name, _, kind, v in verdicts:
        print(f"  {kind:3s} {name:32s} -> {v}")
    dump("wire_verdicts.bin", pickle.dumps(verdicts, protocol=4))


if __name__ == "__main__":
    main()
