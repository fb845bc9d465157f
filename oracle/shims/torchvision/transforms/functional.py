def resize(x, *a, **k): return x
