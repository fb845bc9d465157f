"""Stub: `TensorType[...]` must be subscriptable; annotations only."""
class _TT:
    def __getitem__(self, item):
        return self
    def __call__(self, *a, **k):
        return self
TensorType = _TT()
def patch_typeguard():
    pass
