"""Import the read-only reference (/root/reference) with throw-away shims.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/gen_golden.py` and by the
`tests/test_oracle_vs_reference.py` live cross-checks, which are skipped when
/root/reference is absent (it never exists on the GPU box).  Nothing in the
product package may import this module.  Recipe: SURVEY.md Appendix A.
"""
import os
import sys
import types

REF_ROOT = "/root/reference/nerfstudio"
SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "nerfstudio"))


def import_reference():
    """Returns the imported `nerfstudio` package of the reference, patched for CPU-only use."""
    if not reference_available():
        raise RuntimeError("reference not present at " + REF_ROOT)
    for p in (REF_ROOT, SHIMS):
        if p not in sys.path:
            sys.path.insert(0, p)
    # NS/utils/writer.py:29 imports torch.utils.tensorboard which hard-fails without tensorboard.
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:  # noqa: D401
            def __init__(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb
    import nerfstudio  # noqa: F401
    import nerfstudio.cameras.rays as rays
    import nerfstudio.fields.kplanes_field as kf

    # Latent defect in the reference: kplanes_field.py:422 uses Frustums without importing it.
    kf.Frustums = rays.Frustums
    import nerfstudio.models.kplanes as km

    km.DynMetric = lambda *a, **k: None  # would construct a RetinaNet (dynmetric.py:42-44)
    return nerfstudio
