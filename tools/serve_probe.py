"""Phase timing of one full-size request through the launcher's worker (random-init weights).
usage: python tools/serve_probe.py [n_requests] [candidates]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import serve_worker  # noqa: E402


def main():
    nreq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    args = serve_worker.arg_parser().parse_args(["--random_weights", "--num_inference_steps", "2", "--port", "1"])
    w = serve_worker.build_worker(args)
    torch.cuda.synchronize()
    print(f"[probe] build_worker {time.time() - t0:.1f}s", flush=True)
    fe, den = w.frontend, w.denoise_fn
    enc, dec = fe.encode, fe.decode

    def timed(name, fn):
        def g(*a, **k):
            torch.cuda.synchronize(); t = time.time()
            r = fn(*a, **k)
            torch.cuda.synchronize(); print(f"[probe]   {name}: {time.time() - t:.2f}s", flush=True)
            return r
        return g
    fe.encode, fe.decode, w.denoise_fn = timed("frontend.encode", enc), timed("frontend.decode", dec), timed("denoise(2 steps)", den)
    rs = np.random.RandomState(0)
    for i in range(nreq):
        req = {"b_action": np.array([[4] + [1] * 13] * B, dtype=np.int64), "save_dirs": [f"/tmp/p{i}_{j}" for j in range(B)],
               "request_model_name": "igen", "b_image": rs.randint(0, 256, size=(B, 3, 576, 1024), dtype=np.uint8),
               "return_objects": [True] * B}
        t = time.time()
        out = w(req)
        print(f"[probe] request {i}: {time.time() - t:.2f}s  pred_frames {out['pred_frames'].shape}", flush=True)


if __name__ == "__main__":
    main()
