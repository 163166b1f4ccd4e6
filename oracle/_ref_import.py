"""Import shims for the *reference* (World-In-World) Python path.

TEST INFRASTRUCTURE ONLY.  Used exclusively by oracle/make_golden.py, in the
build container where /root/reference exists, to generate golden vectors and
to validate the oracle restatement.  Nothing here travels to the GPU box as a
dependency of the product path.  Recipe = SURVEY.md Appendix C.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("WIW_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "FTsvd", "diffusers-private", "diffusers"))


def import_reference():
    """Returns a namespace with the reference classes/functions on the SVD hot path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    sys.dont_write_bytecode = True  # /root/reference is read-only
    for p in (os.path.join(REF_ROOT, "FTsvd", "diffusers-private"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import transformers.utils as tu

    if not hasattr(tu, "FLAX_WEIGHTS_NAME"):  # shim 1: removed in transformers 5.x
        tu.FLAX_WEIGHTS_NAME = "flax_model.msgpack"
    import transformers  # noqa: F401  (must be imported before the stubs below)
    from diffusers import (  # type: ignore
        EulerDiscreteScheduler,
        UNetSpatioTemporalConditionModel,
    )

    if "jaxtyping" not in sys.modules:  # shim 2: annotation-only dependency
        jt = types.ModuleType("jaxtyping")

        class _S:
            def __class_getitem__(cls, k):
                return cls

        for n in ("Float", "Int32", "UInt8", "Int", "Bool", "Int64"):
            setattr(jt, n, _S)
        sys.modules["jaxtyping"] = jt
    if "torchvision" not in sys.modules:  # shim 3: one unused helper imports it
        tv = types.ModuleType("torchvision")
        tvt = types.ModuleType("torchvision.transforms")
        tv.transforms = tvt
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tvt
    from utils.svd_utils import get_action_ids  # type: ignore

    ns = types.SimpleNamespace()
    ns.UNet = UNetSpatioTemporalConditionModel
    ns.EulerDiscreteScheduler = EulerDiscreteScheduler
    ns.get_action_ids = get_action_ids
    import diffusers.pipelines.stable_video_diffusion.pipeline_stable_video_diffusion as pl  # type: ignore

    ns.pipeline_module = pl
    return ns
