"""Arithmetic of the multi-GPU gradient / parameter exchange of the K-Planes trainer, free of tensors and devices so that the
plan a SCALE run will execute can be checked on any host (tests/test_exchange_plan_cpu.py).

What it replaces: torch DDP's bucketed all-reduce of every gradient (NS/pipelines/base_pipeline.py:244-246; NSR/scripts/train.py:124-137 sets up NCCL).
Here the field planes (98 % of the floats) go reduce-scatter -> Adam on this rank's shard -> all-gather, in chunks, finest scale first."""
from typing import Dict, List, Sequence, Tuple


def align4(n: int) -> int:
    return (n + 3) // 4 * 4


def plane_layout(C: int, resolutions: Sequence[Sequence[int]]) -> Tuple[List[List[int]], int]:
    """Float offsets of the planes of a PlaneSet (scale-major; planes in the reference's order XY XZ XT YZ YT ZT, plane (a, b) stored [res[b]][res[a]][C])
    and the set's total floats (NS/fields/kplanes_field.py:47-74)."""
    import itertools

    offs, off = [], 0
    for reso in resolutions:
        o = []
        for ca, cb in itertools.combinations(range(len(reso)), 2):
            o.append(off)
            off += reso[cb] * reso[ca] * C
        offs.append(o)
    return offs, off


def field_segment_pad(n: int, world: int) -> int:
    """The field-plane segment padded so that it splits into `world` equal float4-aligned optimiser shards."""
    q = 4 * world
    return (n + q - 1) // q * q


def exchange_chunks(npad: int, finest_offset: int, world: int, chunks: int, n_scales: int) -> List[Dict[str, int]]:
    """The chunks of the field-plane segment in the order they are exchanged (finest scale first): dict(lo, hi, shard).  Chunk boundaries are multiples of
    4 * world, so every rank's shard [lo + r * shard, lo + (r + 1) * shard) of every chunk is float4-aligned; the boundary is rounded UP so that chunk 0 lies
    wholly inside the finest scale.  chunks = 1 (or a single scale): one exchange."""
    q = 4 * world
    assert npad % q == 0
    cuts = [0, npad]
    if chunks > 1 and n_scales > 1:
        b = (finest_offset + q - 1) // q * q
        if 0 < b < npad:
            cuts = [0, b, npad]
    return [{"lo": lo, "hi": hi, "shard": (hi - lo) // world} for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:])))]


def link_bytes(world: int, n_params: int, npad: int, n_reg_values: int, sharded: bool, grad_transport: str = "fp32", param_transport: str = "fp32") -> Dict[str, float]:
    """Bytes one rank SENDS over the links per optimiser step (= bytes it receives), by collective: a reduce-scatter or an all-gather of n elements moves
    (W - 1) / W * n * element size, an all-reduce twice that (ring or direct: the same per-rank volume).  World 1: nothing."""
    W = world
    if W <= 1:
        return {"total": 0.0}
    f = (W - 1) / W
    if sharded:
        eg = 2 if grad_transport == "bf16" else 4
        ep = 2 if param_transport == "bf16" else 4
        small = n_params - npad
        d = {"reduce_scatter.field": f * npad * eg, "all_gather.field": f * npad * ep, "all_reduce.small_segments": 2 * f * small * 4,
             "all_reduce.flags_and_reg_values": 2 * f * (2 * 4 + n_reg_values * 4)}
    else:
        d = {"all_reduce.flat_gradient": 2 * f * n_params * 4, "all_reduce.flags": 2 * f * 2 * 4}
    d["total"] = float(sum(d.values()))
    return d


def kplanes_segment_sizes(base_res: Sequence[int], multiscale: Sequence[int], feature_dim: int, proposal_resolutions: Sequence[Sequence[int]],
                          proposal_feature_dim: int, mlp_param_counts: Dict[str, int], world: int) -> Dict[str, int]:
    """Segment arithmetic of KPlanesTrainer's flat buffer ([proposal planes, proposal nets] x levels, field planes (padded), sigma net, colour net):
    n_params, the field-plane segment (offset, floats, padded floats), the finest scale's offset inside it.  mlp_param_counts: floats of
    "prop" (one proposal net), "sigma", "color" (tcnn_compat.Network.params.numel())."""
    reso = [[r * m for r in base_res[:3]] + list(base_res[3:]) for m in multiscale]
    offs, n_field = plane_layout(feature_dim, reso)
    off = 0
    for r in proposal_resolutions:
        off += align4(plane_layout(proposal_feature_dim, [list(r)])[1])
        off += align4(mlp_param_counts["prop"])
    n_prop = off
    field_off = off
    npad = field_segment_pad(n_field, world)
    off += npad + align4(mlp_param_counts["sigma"]) + align4(mlp_param_counts["color"])
    return {"n_params": off, "n_proposal_params": n_prop, "field_offset": field_off, "field_floats": n_field, "field_padded": npad,
            "finest_offset": offs[-1][0], "n_scales": len(reso)}
