"""The drop-in boundary replayed against bytes and verdicts PRODUCED BY THE REFERENCE's own
downstream/utils/worker_manager.py (fixtures: tests/golden/wire_*.bin, generated in the build container by
oracle/make_wire_golden.py, which imports that module).  CPU only.

  * frames its `write_framed` / `write_pickled_data` wrote are read back by this repo's protocol.read_framed /
    read_pickled, and this repo's writers produce the very same bytes;
  * its `Batcher.split_batch` sub-batches == plumbing.split_batch; its `_recompose_batch` of this repo's responses
    == plumbing.recompose; the recomposed response is what the client receives (list of b uint8 arrays);
  * its `check_inputdict` / `check_outputdict` verdicts on 17 well- and mal-formed dicts == plumbing's verdicts
    (same exception class), including "the reference ACCEPTS this repo's response".
"""
import io
import os
import pickle
import socket
import threading

import numpy as np
import pytest

import wiw_amd  # noqa: F401
from wiw_amd.server import plumbing as P
from wiw_amd.server import protocol as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blob(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def same(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.dtype == b.dtype and np.array_equal(a, b)
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    return a == b


def feed_socket(data: bytes, chunk=1000):
    """A connected socket whose peer dribbles `data` in small chunks (framing must survive any chunking)."""
    a, b = socket.socketpair()

    def run():
        for i in range(0, len(data), chunk):
            a.sendall(data[i:i + chunk])
        a.close()

    threading.Thread(target=run, daemon=True).start()
    return b


def test_read_reference_written_request_and_done():
    raw = blob("wire_request.bin")
    s = feed_socket(raw + blob("wire_done.bin"))
    req = W.read_framed(s)
    assert set(req) == {"b_action", "save_dirs", "request_model_name", "b_image", "return_objects"}
    assert req["b_action"].dtype == np.int64 and req["b_action"].shape == (3, 14)
    assert req["b_image"].dtype == np.uint8 and req["b_image"].shape == (3, 3, 16, 32)
    P.check_inputdict(req)
    assert W.read_framed(s) == W.DONE
    with pytest.raises(EOFError):
        W.read_framed(s)
    # this repo's writer produces the reference's bytes
    assert W.dumps_frame(req) == raw
    assert W.dumps_frame(W.DONE) == blob("wire_done.bin")


def test_read_reference_written_worker_task_and_result():
    raw = blob("wire_task.bin")
    f = io.BytesIO(raw)
    client_id, task_id, payload = W.read_pickled(f)
    assert (client_id, task_id) == (7, 42) and payload["b_action"].shape == (1, 14) and payload["save_dirs"] == ["/tmp/wiw_wire/cand_1"]
    assert payload["request_model_name"] == "g"        # the manager slices the NAME too (worker_manager.py:458)
    out = io.BytesIO()
    W.write_pickled(out, (client_id, task_id, payload))
    assert out.getvalue() == raw
    cid, tid, result = W.read_pickled(io.BytesIO(blob("wire_result.bin")))
    assert (cid, tid) == (7, 42) and result["pred_frames"].dtype == np.uint8 and result["pred_frames"].shape == (1, 14, 3, 8, 8)
    P.check_outputdict(result)


def test_worker_main_consumes_reference_task_bytes():
    """The manager-compatible loop fed with the reference's stdin bytes answers on the pipe in the reference's framing."""
    from wiw_amd.server.worker import worker_main

    r, w = os.pipe()
    seen = []

    def task_fn(payload):
        seen.append(payload)
        return {"save_dirs": list(payload["save_dirs"]), "pred_frames": np.zeros((1, 14, 3, 8, 8), np.uint8)}

    worker_main(w, task_fn, stdin=io.BytesIO(blob("wire_task.bin") + W.dumps_frame(W.DONE)))
    with os.fdopen(r, "rb") as f:
        cid, tid, res = W.read_pickled(f)
    assert (cid, tid) == (7, 42) and len(seen) == 1 and res["save_dirs"] == ["/tmp/wiw_wire/cand_1"]


@pytest.mark.parametrize("bs", [1, 2, 3])
def test_batcher_split_and_recompose_match_reference(bs):
    req = W.read_framed(feed_socket(blob("wire_request.bin")))
    ref = pickle.loads(blob("wire_batcher.bin"))[bs]
    subs = P.split_batch(req, bs)
    assert len(subs) == len(ref["subs"]) == -(-3 // bs) and ref["next_id"] == len(subs)
    for (tid, rsub), (i, sub) in zip(ref["subs"], enumerate(subs)):
        assert tid == i and same(rsub, sub)
    out = P.recompose([r for _, r in sorted(ref["responses"], key=lambda x: x[0])])
    assert same(out, ref["recomposed"])
    # what the client finally holds behind the manager: a LIST of b uint8 (14,3,h,w) arrays, save_dirs in order
    assert isinstance(out["pred_frames"], list) and len(out["pred_frames"]) == 3 and out["pred_frames"][0].dtype == np.uint8
    assert out["save_dirs"] == req["save_dirs"]


def test_check_verdicts_match_reference():
    cases = pickle.loads(blob("wire_verdicts.bin"))
    assert len(cases) == 17 and {v for *_, v in cases} == {"ok", "KeyError", "AssertionError"}
    for name, d, kind, ref_verdict in cases:
        fn = P.check_inputdict if kind == "in" else P.check_outputdict
        try:
            fn(d)
            got = "ok"
        except Exception as e:
            got = type(e).__name__
        assert got == ref_verdict, f"{name}: reference says {ref_verdict}, this repo says {got}"
