"""Measured parity deviations (bounds must sit near what is measured, and the measurement must be on record).
record(name, got, want) notes max |got - want| and the largest error relative to max(|want|, floor); everything lands in
gpurun_out/parity_deviations.json (scratch; the round's copy is committed under profiles/ and quoted in DESIGN.md section 2)."""
import json
import os

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OUT = os.path.join(_ROOT, "gpurun_out", "parity_deviations.json")
_seen = {}


def record(name: str, got, want, floor: float = 0.0) -> dict:
    got, want = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(want).detach().double().cpu()
    diff = (got - want).abs()
    scale = float(want.abs().max()) if want.numel() else 0.0
    rel_el = float((diff / want.abs().clamp_min(max(floor, 1e-300))).max()) if want.numel() else 0.0
    d = {"max_abs": float(diff.max()) if diff.numel() else 0.0, "max_abs_over_max_want": (float(diff.max()) / scale) if scale else 0.0,
         "max_rel_elementwise": rel_el, "floor": floor, "want_abs_max": scale, "numel": int(want.numel())}
    prev = _seen.get(name)
    if prev is None or d["max_abs"] > prev["max_abs"]:
        _seen[name] = d
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        old = json.load(open(_OUT)) if os.path.exists(_OUT) else {}
        old.update(_seen)
        json.dump(old, open(_OUT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    return d
