"""The multi-GPU exchange plan of the K-Planes trainer at the PRESET's sizes (BASELINE config 5: 8 x MI355X), checked on the CPU: no SCALE record exists
yet (the driver has had no multi-GPU node), so what such a run will put on the links is pinned here from the same arithmetic the trainer executes
(soccernerfs_amd/exchange_plan.py; KPlanesTrainer asserts at construction that this arithmetic describes its flat buffer).

Reference side: torch DDP all-reduces every gradient once per step (NS/pipelines/base_pipeline.py:244-246): 2 (W-1)/W x 156 049 664 x 4 B per rank.  Here
the field planes go reduce-scatter -> shard Adam -> all-gather (the same bytes as that all-reduce), the 2.9 M floats of the small segments one all-reduce."""
import pytest

from soccernerfs_amd import exchange_plan as XP

PRESET = dict(base_res=(64, 64, 64, 100), multiscale=(1, 2, 4, 8, 16), feature_dim=32, proposal_resolutions=((128, 128, 128, 100), (256, 256, 256, 100)),
              proposal_feature_dim=8)
REG_VALUES = 64 * 16  # ops.REG_SLOTS x 16 floats of the field planes' regulariser sums


def _mlp_counts():
    from soccernerfs_amd.tcnn_compat import Network

    mk = lambda din, dout, h, nh, act: Network(din, dout, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": h,
                                                           "n_hidden_layers": nh}).params.numel()
    return {"prop": mk(8, 1, 64, 1, "None"), "sigma": mk(160, 16, 128, 1, "None"), "color": mk(15, 3, 64, 2, "Sigmoid")}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_preset_exchange_plan(world):
    sz = XP.kplanes_segment_sizes(PRESET["base_res"], PRESET["multiscale"], PRESET["feature_dim"], PRESET["proposal_resolutions"], PRESET["proposal_feature_dim"],
                                  _mlp_counts(), world)
    assert sz["n_params"] == 156_049_664 and sz["field_floats"] == 153_133_056  # BASELINE config 2's parameter count; DESIGN section 4's sweep size
    npad = sz["field_padded"]
    assert npad % (4 * world) == 0 and 0 <= npad - sz["field_floats"] < 4 * world
    chunks = XP.exchange_chunks(npad, sz["finest_offset"], world, 2, sz["n_scales"])
    assert len(chunks) == 2
    # finest scale first, and chunk 0 lies wholly inside it (72 % of the plane floats)
    assert chunks[0]["hi"] == npad and chunks[0]["lo"] >= sz["finest_offset"] and chunks[0]["lo"] - sz["finest_offset"] < 4 * world
    assert chunks[1]["lo"] == 0 and chunks[1]["hi"] == chunks[0]["lo"]
    assert 0.71 < (chunks[0]["hi"] - chunks[0]["lo"]) / npad < 0.73
    covered = 0
    for ch in chunks:
        n = ch["hi"] - ch["lo"]
        assert ch["lo"] % (4 * world) == 0 and n % (4 * world) == 0 and ch["shard"] * world == n and ch["shard"] % 4 == 0
        # the shards of the ranks tile the chunk exactly once
        edges = [ch["lo"] + r * ch["shard"] for r in range(world + 1)]
        assert edges[-1] == ch["hi"] and all(b - a == ch["shard"] for a, b in zip(edges[:-1], edges[1:]))
        covered += n
    assert covered == npad
    # a single exchange on request, or for a single-scale model
    assert XP.exchange_chunks(npad, sz["finest_offset"], world, 1, sz["n_scales"]) == [{"lo": 0, "hi": npad, "shard": npad // world}]
    assert XP.exchange_chunks(npad, 0, world, 2, 1) == [{"lo": 0, "hi": npad, "shard": npad // world}]

    f = (world - 1) / world
    lb = XP.link_bytes(world, sz["n_params"], npad, REG_VALUES, sharded=True)
    assert lb["reduce_scatter.field"] == lb["all_gather.field"] == f * npad * 4
    small = sz["n_params"] - npad
    assert small == 2_916_608 and lb["all_reduce.small_segments"] == 2 * f * small * 4
    assert lb["total"] == pytest.approx(sum(v for k, v in lb.items() if k != "total"))
    # the same volume as DDP's one all-reduce of every gradient (the reference), to within the flags
    ddp = XP.link_bytes(world, sz["n_params"], npad, REG_VALUES, sharded=False)
    assert abs(lb["total"] - ddp["total"]) < 1e4 and ddp["all_reduce.flat_gradient"] == 2 * f * sz["n_params"] * 4
    # bf16 transports halve the two big collectives, nothing else
    h = XP.link_bytes(world, sz["n_params"], npad, REG_VALUES, sharded=True, grad_transport="bf16", param_transport="bf16")
    assert h["reduce_scatter.field"] == lb["reduce_scatter.field"] / 2 and h["all_gather.field"] == lb["all_gather.field"] / 2
    assert h["all_reduce.small_segments"] == lb["all_reduce.small_segments"]
    if world == 8:
        # DESIGN section 6's figures: 536 + 536 + 20 MB per rank and step
        assert round(lb["reduce_scatter.field"] / 1e6) == 536 and round(lb["all_gather.field"] / 1e6) == 536 and round(lb["all_reduce.small_segments"] / 1e6) == 20
        assert round(lb["total"] / (7 * 153e9) * 1e3, 2) == 1.02  # ms with all seven xGMI links of a GPU busy


def test_world_one_moves_nothing_and_plane_layout_matches_planeset():
    assert XP.link_bytes(1, 10, 8, 4, True) == {"total": 0.0}
    from soccernerfs_amd.plane_set import PlaneSet

    reso = [[4, 5, 6, 3], [8, 10, 12, 3]]
    ps = PlaneSet(8, reso, concat=True)
    offs, n = XP.plane_layout(8, reso)
    assert offs == ps.offsets and n == ps.numel
