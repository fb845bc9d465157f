def imread(*a, **k): raise RuntimeError("imageio stub")
v2 = None
