"""nerfstudio checkpoint files: `step-%09d.ckpt` as NS/engine/trainer.py:353-374 writes them and :331-351 reads them back --
torch.save({"step", "pipeline": {"_model.<name>": tensor, ...}, "optimizers": {group: Adam.state_dict()}, "scalers": ...}).

The tensors are converted between this package's layouts and the reference's names / shapes:
  planes        flat channel-last buffer (plane_set.PlaneSet)      <->  `grids.{s}.{p}` [1,C,H,W] (field: nested by scale; proposal
                                                                         networks: a bare list `grids.{p}`)
  tiny MLPs     one flat input-major vector (tcnn_compat.Network)   <->  `layers.{k}.weight` [out,in] (the fp32 Linear-stack naming of the
                                                                         reference run without tiny-cuda-nn; real tcnn stores one
                                                                         opaque `params` vector per net, whose padded fp16 layout is
                                                                         tcnn's own and is not imported here)
  hash tables   `embeddings` / `params`                             <->  same names, same shapes (+ the temporal grid's index buffers on
                                                                         export, so the reference's strict load_state_dict accepts it)
  nn.Embedding  `weight`                                            <->  `embedding.weight` (NS/field_components/embedding.py wrapper)
Adam moments go through the same (linear) conversions, so a reference optimiser state resumes here and vice versa.
"""
import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import nn

from .plane_set import PlaneSet
from .tcnn_compat import Network
from .temporal_grid import TemporalGridEncoder


class _Entry:
    """One of this package's parameters and the reference tensors it corresponds to."""

    def __init__(self, name: str, param: torch.Tensor, keys: List[str], to_ref: Callable, from_ref: Callable):
        self.name, self.param, self.keys, self.to_ref, self.from_ref = name, param, keys, to_ref, from_ref


def _entries(model: nn.Module) -> List[_Entry]:
    out: List[_Entry] = []
    claimed = set()
    for path, mod in model.named_modules():
        pre = path + "." if path else ""
        if isinstance(mod, PlaneSet):
            nested = not path.startswith("proposal_networks")  # KPlanesField nests grids by scale, KPlanesDensityField does not
            keys = [(f"{pre}{s}.{p}" if nested else f"{pre}{p}") for s in range(len(mod.resolutions)) for p in range(len(mod.combs))]

            def to_ref(buf, m=mod):
                return [t for scale in m.to_reference(buf) for t in scale]

            def from_ref(ts, m=mod):
                buf = torch.empty(m.numel, dtype=torch.float32)
                i = 0
                for s in range(len(m.resolutions)):
                    for p in range(len(m.combs)):
                        m.plane_view(s, p, buf).copy_(ts[i][0].permute(1, 2, 0))
                        i += 1
                return buf

            out.append(_Entry(pre + "planes", mod.planes, keys, to_ref, from_ref))
            claimed.add(pre + "planes")
        elif isinstance(mod, Network):
            keys = [f"{pre}layers.{i}.weight" for i in range(len(mod.dims) - 1)]

            def to_ref(buf, m=mod):
                return m.linear_weights(buf)

            def from_ref(ts, m=mod):
                return torch.cat([w.t().reshape(-1) for w in ts])

            out.append(_Entry(pre + "params", mod.params, keys, to_ref, from_ref))
            claimed.add(pre + "params")
        elif isinstance(mod, nn.Embedding):
            out.append(_Entry(pre + "weight", mod.weight, [pre + "embedding.weight"], lambda b: [b.detach().clone()], lambda ts: ts[0]))
            claimed.add(pre + "weight")
    for name, p in model.named_parameters():
        if name not in claimed:
            out.append(_Entry(name, p, [name], lambda b: [b.detach().clone()], lambda ts: ts[0]))
    return out


def _buffers(model: nn.Module) -> "OrderedDict[str, torch.Tensor]":
    """Index buffers the reference's TemporalGridEncoder registers (temporal_grid.py:226,304-308)."""
    out = OrderedDict()
    for path, mod in model.named_modules():
        if isinstance(mod, TemporalGridEncoder):
            for b in ("offsets", "index_list", "sampling_index", "index_a_mask", "index_b_mask"):
                out[f"{path}.{b}"] = getattr(mod, b).detach().cpu().clone()
    return out


def reference_state_dict(model: nn.Module, prefix: str = "_model.") -> "OrderedDict[str, torch.Tensor]":
    """The model's parameters under the reference's state_dict names (pipeline prefix `_model.`, base_pipeline.py:109-113)."""
    sd = OrderedDict()
    for e in _entries(model):
        for k, t in zip(e.keys, e.to_ref(e.param.detach())):
            sd[prefix + k] = t.detach().cpu().contiguous()
    for k, t in _buffers(model).items():
        sd[prefix + k] = t
    return sd


@torch.no_grad()
def load_reference_state_dict(model: nn.Module, state: Dict[str, torch.Tensor], prefix: str = "_model.", strict: bool = True) -> None:
    used = set()
    for e in _entries(model):
        ks = [prefix + k for k in e.keys]
        missing = [k for k in ks if k not in state]
        if missing:
            if strict:
                raise KeyError(f"checkpoint is missing {missing[:3]}{'...' if len(missing) > 3 else ''} (for {e.name})")
            continue
        t = e.from_ref([state[k].float() for k in ks])
        if t.numel() != e.param.numel():
            raise RuntimeError(f"{e.name}: checkpoint holds {t.numel()} values, the model {e.param.numel()}")
        e.param.copy_(t.reshape(e.param.shape).to(e.param.device))
        used.update(ks)
    if strict:
        extra = [k for k in state if k.startswith(prefix) and k not in used and k[len(prefix):] not in _buffers(model)]
        if extra:
            raise KeyError(f"unexpected keys in checkpoint: {extra[:5]}")


def _group_of(key: str) -> Optional[str]:
    """Parameter group of a reference key (get_param_groups, kplanes.py:311-316 / nerfacto.py:229-233)."""
    if key.startswith("proposal_networks."):
        return "proposal_networks"
    if key.startswith("field."):
        return "fields"
    return None


def _group_layout(model: nn.Module) -> Dict[str, List[Tuple[_Entry, List[int]]]]:
    """For each optimiser group: (entry, indices of its reference tensors in the group's torch parameter order = module registration
    order, i.e. the order of the state_dict keys)."""
    groups: Dict[str, List[Tuple[_Entry, List[int]]]] = {"proposal_networks": [], "fields": []}
    counters = {"proposal_networks": 0, "fields": 0}
    for e in _entries_in_reference_order(model):
        g = _group_of(e.keys[0])
        if g is None:
            continue
        idx = list(range(counters[g], counters[g] + len(e.keys)))
        counters[g] += len(e.keys)
        groups[g].append((e, idx))
    return groups


def _entries_in_reference_order(model: nn.Module) -> List[_Entry]:
    """Reference registration order inside a field: aabb first, then encodings / embeddings / MLPs in construction order -- the same
    order this package's mirrors construct them in, except that wrapped parameters were collected before the plain ones above."""
    ents = _entries(model)
    order = {name: i for i, (name, _) in enumerate(model.named_parameters())}
    return sorted(ents, key=lambda e: order[e.name])


def export_optimizer_states(model: nn.Module, moments: Dict[str, Tuple[torch.Tensor, torch.Tensor, int]], hyper: Dict[str, Dict]) -> Dict[str, Dict]:
    """moments: this package's parameter name -> (exp_avg, exp_avg_sq, step) in the parameter's own layout.
    -> {group: torch.optim.Adam-style state_dict} in the reference's parameter order."""
    out = {}
    for g, items in _group_layout(model).items():
        state, n = {}, 0
        for e, idx in items:
            n = max(n, idx[-1] + 1)
            if e.name not in moments:
                continue
            m, v, step = moments[e.name]
            for i, tm, tv in zip(idx, e.to_ref(m.detach()), e.to_ref(v.detach())):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": tm.cpu().contiguous(), "exp_avg_sq": tv.cpu().contiguous()}
        out[g] = {"state": state, "param_groups": [{**hyper.get(g, {}), "params": list(range(n))}]}
    return out


def import_optimizer_states(model: nn.Module, optimizers: Dict[str, Dict]) -> Dict[str, Tuple[torch.Tensor, torch.Tensor, int]]:
    """Inverse of export_optimizer_states: -> parameter name -> (exp_avg, exp_avg_sq, step) in this package's layouts (parameters the
    reference optimiser never stepped are absent)."""
    out = {}
    for g, items in _group_layout(model).items():
        st = optimizers.get(g, {}).get("state", {})
        for e, idx in items:
            if not all(i in st for i in idx):
                continue
            m = e.from_ref([st[i]["exp_avg"].float() for i in idx]).reshape(e.param.shape)
            v = e.from_ref([st[i]["exp_avg_sq"].float() for i in idx]).reshape(e.param.shape)
            out[e.name] = (m, v, int(st[idx[0]]["step"]))
    return out


def checkpoint_path(checkpoint_dir: str, step: int) -> str:
    return os.path.join(checkpoint_dir, f"step-{step:09d}.ckpt")


def save_checkpoint(checkpoint_dir: str, step: int, model: nn.Module, optimizers: Optional[Dict[str, Dict]] = None,
                    save_only_latest_checkpoint: bool = True) -> str:
    """trainer.py:353-380."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = checkpoint_path(checkpoint_dir, step)
    torch.save({"step": step, "pipeline": reference_state_dict(model), "optimizers": optimizers or {}, "scalers": {}}, path)
    if save_only_latest_checkpoint:
        for f in os.listdir(checkpoint_dir):
            if f.endswith(".ckpt") and os.path.join(checkpoint_dir, f) != path:
                os.unlink(os.path.join(checkpoint_dir, f))
    return path


def load_checkpoint(load_dir: str, model: nn.Module, load_step: Optional[int] = None, strict: bool = True):
    """trainer.py:331-351: the latest `step-*.ckpt` of load_dir unless load_step is given.  -> (start step = saved step + 1, optimiser
    moments per parameter as import_optimizer_states returns them)."""
    if load_step is None:
        steps = sorted(int(f[f.find("-") + 1: f.find(".")]) for f in os.listdir(load_dir) if f.startswith("step-") and f.endswith(".ckpt"))
        if not steps:
            raise FileNotFoundError(f"no step-*.ckpt in {load_dir}")
        load_step = steps[-1]
    path = checkpoint_path(load_dir, load_step)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint {path} does not exist")
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    load_reference_state_dict(model, loaded["pipeline"], strict=strict)
    return loaded["step"] + 1, import_optimizer_states(model, loaded.get("optimizers", {}))
