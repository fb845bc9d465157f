"""Host arithmetic of the tiled temporal-grid backward (snerf_tgrid_tile_plan_make, csrc/tgrid_tiles.hip): no GPU needed."""
import ctypes as C


def test_plan_of_the_preset_tables():
    """The tiling arithmetic at config 4's sizes (host only): 256-row tiles of the 66-column main table (two workgroups per CU), 512-row tiles of the
    34-column proposal tables; levels below 2^16 rows stay atomic."""
    from soccernerfs_amd import _lib
    from soccernerfs_amd.temporal_grid import TemporalGridEncoder

    enc = TemporalGridEncoder(input_dim=3, temporal_dim=64, num_levels=16, level_dim=2, log2_hashmap_size=19, desired_resolution=2048)
    plan = _lib.TgridTilePlan()
    _lib.check(_lib.lib().snerf_tgrid_tile_plan_make(C.byref(enc.desc), C.c_int64(4096 * 48), 0, -1, C.byref(plan)))
    assert plan.tile_rows_log2 == 8 and plan.lds_bytes <= 72 * 1024 and plan.first_tiled_level == 3
    offs = enc.offsets.tolist()
    assert plan.n_tiles == sum(-(-(offs[l + 1] - offs[l]) // 256) for l in range(16))
    assert plan.n_chunks == 48 and plan.record_capacity == 4096 * 48 * 13 * 8
    for args, sh, lc in (({"max_res": 64}, 9, 3), ({"max_res": 256}, 9, 2)):
        import numpy as np

        growth = float(np.exp((np.log(args["max_res"]) - np.log(16)) / 4))
        pe = TemporalGridEncoder(input_dim=3, temporal_dim=32, num_levels=5, level_dim=2, per_level_scale=growth, base_resolution=16, log2_hashmap_size=17)
        _lib.check(_lib.lib().snerf_tgrid_tile_plan_make(C.byref(pe.desc), C.c_int64(4096 * 256), 0, -1, C.byref(plan)))
        assert plan.tile_rows_log2 == sh and plan.first_tiled_level == lc and plan.lds_bytes <= 72 * 1024, (plan.tile_rows_log2, plan.first_tiled_level)
        assert plan.n_chunks == 256


def test_plan_of_the_preset_hash_table():
    """snerf_hashgrid_tile_plan_make at the full NeRFPlayer's static hash grid (16 levels, F = 2, 2^19 rows per hashed level): 2^11-row tiles, the levels below
    2^18 rows (0-3: 4096 ... 110 592 rows, where every point of the batch lands in a handful of tiles) stay atomic."""
    from soccernerfs_amd import _lib
    from soccernerfs_amd.tcnn_compat import Encoding

    enc = Encoding(3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 1.4472692012786865})
    plan = _lib.HashgridTilePlan()
    B = 2 * 4096 * 48
    _lib.check(_lib.lib().snerf_hashgrid_tile_plan_make(C.byref(enc.desc), C.c_int64(B), 0, -1, C.byref(plan)))
    offs = [enc.desc.offsets[l] for l in range(17)]
    assert plan.tile_rows_log2 == 11 and plan.lds_bytes == 2048 * 2 * 4
    assert plan.first_tiled_level == next(l for l in range(16) if offs[l + 1] - offs[l] >= 1 << 18)
    assert plan.n_tiles == sum(-(-(offs[l + 1] - offs[l]) // 2048) for l in range(16))
    assert plan.record_capacity == B * (16 - plan.first_tiled_level) * 8 and plan.n_chunks == B // 4096
