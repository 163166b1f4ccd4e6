"""WM-server boundary: the wire protocol, request/response plumbing and worker loops of the reference
(downstream/utils/worker_manager.py, downstream/api_models/__init__.py, FTsvd/eval_inference.py)
re-implemented so the HIP denoiser is a drop-in world-model backend for `downstream/solver_*`."""
