class LearnedPerceptualImagePatchSimilarity:
    def __init__(self, *a, **k): pass
    def __call__(self, *a, **k): raise RuntimeError("torchmetrics stub")
    def to(self, *a, **k): return self
