class _T:
    def __init__(self, *a, **k): pass
    def __call__(self, x): return x
Compose = Resize = ToTensor = Normalize = _T
from . import functional  # noqa
