"""ctypes binding of libsnerf.so (C ABI: include/snerf.h).  The product path has NO fallback:
if the HIP library is missing or stale the import fails loudly."""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: libsnerf must bind to the HIP runtime torch already loaded (same SONAME)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libsnerf.so")

MAX_SCALES = 8
ABI_VERSION = 14


class KPlanesDesc(C.Structure):
    _fields_ = [
        ("n_scales", C.c_int32),
        ("C", C.c_int32),
        ("concat", C.c_int32),
        ("n_coords", C.c_int32),
        ("res", (C.c_int32 * 4) * MAX_SCALES),
        ("off", (C.c_int64 * 6) * MAX_SCALES),
    ]


class Coords(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("S", C.c_int32),
        ("rescale", C.c_int32),
        ("_pad", C.c_int32),
        ("pts", C.c_void_p),
        ("origins", C.c_void_p),
        ("dirs", C.c_void_p),
        ("times", C.c_void_p),
        ("ebins", C.c_void_p),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
    ]


class ResampleArgs(C.Structure):
    _fields_ = [
        ("density", C.c_void_p), ("weights_in", C.c_void_p), ("ebins_prev", C.c_void_p), ("weights_out", C.c_void_p),
        ("sbins_prev", C.c_void_p), ("u_or_rand", C.c_void_p), ("nears", C.c_void_p), ("fars", C.c_void_p),
        ("sbins_out", C.c_void_p), ("ebins_out", C.c_void_p), ("inds_out", C.c_void_p),
        ("R", C.c_int32), ("S_prev", C.c_int32), ("S", C.c_int32),
        ("u_mode", C.c_int32), ("rand_cols", C.c_int32), ("kind", C.c_int32),
        ("anneal", C.c_float), ("histogram_padding", C.c_float), ("eps", C.c_float),
    ]


class MlpDesc(C.Structure):
    _fields_ = [("d_in", C.c_int32), ("hidden", C.c_int32), ("n_hidden", C.c_int32), ("d_out", C.c_int32),
                ("hidden_act", C.c_int32), ("out_act", C.c_int32), ("operands", C.c_int32)]


class AdamDyn(C.Structure):
    """snerf_adam_dyn: device-resident optimiser state of one parameter group (8 x 4 bytes; lives in a torch int32[8] tensor)."""
    _fields_ = [("nonfinite", C.c_int32), ("t", C.c_int32), ("skipped", C.c_int32), ("dropped", C.c_int32),
                ("step_size", C.c_float), ("inv_sqrt_bc2", C.c_float), ("skip", C.c_int32), ("_pad", C.c_int32)]


class RenderArgs(C.Structure):
    _fields_ = [("weights", C.c_void_p), ("rgb", C.c_void_p), ("ebins", C.c_void_p), ("bg", C.c_void_p),
                ("R", C.c_int32), ("S", C.c_int32), ("bg_mode", C.c_int32), ("training", C.c_int32),
                ("rgb_out", C.c_void_p), ("acc_out", C.c_void_p), ("depth_median", C.c_void_p), ("depth_expected", C.c_void_p),
                ("median_rgb", C.c_void_p), ("median_index", C.c_void_p)]


class RayTrainArgs(C.Structure):
    _fields_ = [("density", C.c_void_p), ("ebins", C.c_void_p), ("sbins", C.c_void_p), ("rgb", C.c_void_p), ("bg", C.c_void_p), ("target", C.c_void_p),
                ("R", C.c_int32), ("S", C.c_int32), ("bg_mode", C.c_int32), ("go_scale", C.c_float), ("dist_scale", C.c_float),
                ("weights", C.c_void_p), ("rgb_out", C.c_void_p), ("acc_out", C.c_void_p), ("depth_median", C.c_void_p), ("sqerr_rays", C.c_void_p),
                ("dist_rays", C.c_void_p), ("g_rgb", C.c_void_p), ("g_density", C.c_void_p), ("g_weights", C.c_void_p), ("nonfinite_flag", C.c_void_p)]


class RaygenArgs(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("fx", C.c_void_p), ("fy", C.c_void_p), ("cx", C.c_void_p), ("cy", C.c_void_p),
                ("c2w", C.c_void_p), ("cam_times", C.c_void_p),
                ("R", C.c_int32), ("collide", C.c_int32), ("training", C.c_int32), ("near_plane", C.c_float),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
                ("origins", C.c_void_p), ("dirs", C.c_void_p), ("pixel_area", C.c_void_p), ("dir_norm", C.c_void_p),
                ("times", C.c_void_p), ("nears", C.c_void_p), ("fars", C.c_void_p)]


class TgridDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("grid_C", C.c_int32), ("H", C.c_int32), ("gridtype", C.c_int32),
                ("align_corners", C.c_int32), ("S", C.c_float), ("offsets", C.c_int32 * 33)]


class TgridTilePlan(C.Structure):
    """snerf_tgrid_tile_plan (ABI 14): the tiling of a temporal-grid table for the owner-computes backward (csrc/tgrid_tiles.hip)."""
    _fields_ = [("tile_rows_log2", C.c_int32), ("n_tiles", C.c_int32), ("n_chunks", C.c_int32), ("chunk", C.c_int32), ("first_tiled_level", C.c_int32),
                ("lds_bytes", C.c_int32), ("tile_start", C.c_int32 * 33), ("_pad", C.c_int32), ("count_ints", C.c_int64), ("record_capacity", C.c_int64)]


class HashgridTilePlan(C.Structure):
    """snerf_hashgrid_tile_plan (ABI 14, csrc/hashgrid_tiles.hip)."""
    _fields_ = [("tile_rows_log2", C.c_int32), ("n_tiles", C.c_int32), ("n_chunks", C.c_int32), ("chunk", C.c_int32), ("lds_bytes", C.c_int32), ("first_tiled_level", C.c_int32),
                ("tile_start", C.c_int32 * 33), ("_pad2", C.c_int32), ("count_ints", C.c_int64), ("record_capacity", C.c_int64)]


class HashgridDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("F", C.c_int32), ("L", C.c_int32), ("scale", C.c_float * 32), ("resolution", C.c_int32 * 32),
                ("offsets", C.c_int32 * 33)]


_lib = None


def lib():
    """Returns the loaded library, raising a RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP library is the product path and has no fallback. "
            "Build it with `python -m soccernerfs_amd.build` (or __graft_entry__.build())."
        )
    l = C.CDLL(LIB_PATH)
    l.snerf_last_error.restype = C.c_char_p
    l.snerf_target_arch.restype = C.c_char_p
    l.snerf_mlp_param_count.restype = C.c_int64
    l.snerf_mlp_gw_workspace_floats.restype = C.c_int64
    l.snerf_hashgrid_layout.restype = C.c_int64
    l.snerf_hashgrid_layout.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_int32]
    # float arguments must be declared or ctypes passes them as ints/doubles
    F, I, L, P = C.c_float, C.c_int32, C.c_int64, C.c_void_p
    l.snerf_distortion.argtypes = [P, P, I, I, F, P, P, I, P]
    l.snerf_depth_loss.argtypes = [P, P, P, P, F, I, I, F, P, P, I, P]
    l.snerf_urf_depth_loss.argtypes = [P, P, P, P, P, F, I, I, F, P, P, P, I, P]
    l.snerf_interlevel.argtypes = [P, P, I, P, P, I, I, F, P, P, P]
    l.snerf_plane_reg.argtypes = [P, P, P, F, F, F, P, I, I, P]
    l.snerf_adam_step.argtypes = [P, P, P, P, P, L, F, F, F, F, I, F, I, P, P]
    l.snerf_adam_planes_step.argtypes = [P, P, P, P, P, P, F, F, F, P, I, F, F, F, F, I, F, I, P, P]
    l.snerf_adam_planes_step_range.argtypes = [P, P, P, P, P, P, F, F, F, P, I, F, F, F, F, I, F, I, L, L, P, P]
    l.snerf_adam_prepare.argtypes = [P, F, F, F, I, I, P]
    l.snerf_weights_bwd.argtypes = [P, P, P, I, I, P, I, P, P]
    l.snerf_fx_to_float.argtypes = [P, P, L, I, P]
    l.snerf_aabb_collide.argtypes = [P, P, I, P, F, I, P, P, P]
    l.snerf_render_bwd.argtypes = [P, P, P, I, P, P, I, I, P, P, I, P]
    l.snerf_render_mse_bwd.argtypes = [P, P, P, I, P, P, F, I, I, P, P, P, P]
    l.snerf_sample_pixels_uniform.argtypes = [P, I, I, I, I, P, P, P, P]
    l.snerf_sort_rays_by_key.argtypes = [P, P, I, I, P, I, P, P, P]
    l.snerf_trunc_exp_fwd.argtypes = [P, L, P, P]
    l.snerf_trunc_exp_bwd.argtypes = [P, P, L, P, P]
    l.snerf_basis_rgb_fwd.argtypes = [P, I, P, L, I, P, P]
    l.snerf_basis_rgb_bwd.argtypes = [P, I, P, P, P, L, I, P, P, P]
    l.snerf_tgrid_tv_fwd.argtypes = [P, L, I, I, I, P, I, P]
    l.snerf_tgrid_tv_bwd.argtypes = [P, L, I, I, I, P, P, P]
    l.snerf_tgrid_tv_fwd_bwd.argtypes = [P, L, I, I, I, F, P, I, P, P]
    l.snerf_tgrid_tv_sign.argtypes = [P, L, I, I, I, F, P, I, P, P]
    l.snerf_tgrid_tile_plan_make.argtypes = [P, L, I, I, P]
    l.snerf_tgrid_bwd_bin.argtypes = [P, P, P, P, I, L, P, P, P, P, P, P]
    l.snerf_tgrid_bwd_tiles.argtypes = [P, P, L, P, P, P, P, P, P]
    l.snerf_tgrid_bwd_tiles_adam.argtypes = [P, P, L, P, P, P, P, P, P, P, P, F, F, F, F, I, I, I, P, P]
    l.snerf_tgrid_encode_bwd_levels.argtypes = [P, P, P, P, I, L, P, P, I, I, P]
    l.snerf_hashgrid_tile_plan_make.argtypes = [P, L, I, I, P]
    l.snerf_hashgrid_encode_bwd_levels.argtypes = [P, P, P, L, P, P, P, I, I, P]
    l.snerf_hashgrid_bwd_bin.argtypes = [P, P, P, L, P, P, P, P, P]
    l.snerf_hashgrid_bwd_tiles.argtypes = [P, P, P, L, P, P, P, P, P]
    l.snerf_hashgrid_bwd_tiles_adam.argtypes = [P, P, P, L, P, P, P, P, P, P, P, F, F, F, F, I, P]
    l.snerf_adam_step_tv.argtypes = [P, P, P, P, L, I, I, I, P, F, F, F, F, I, F, I, P, P]
    l.snerf_isg_maps.argtypes = [P, I, I, I, I, I, P, P, P, I, F, P, P, P]
    l.snerf_ist_maps.argtypes = [P, I, I, I, I, P, P, F, P, P]
    if l.snerf_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libsnerf ABI {l.snerf_abi_version()} != binding {ABI_VERSION}: rebuild the library")
    _lib = l
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().snerf_last_error().decode()
        raise RuntimeError(f"libsnerf {what} failed (code {rc}): {msg}")


# every symbol include/snerf.h declares; tests/test_abi.py checks the list against the header
EXPORTS = [
    "snerf_abi_version",
    "snerf_last_error",
    "snerf_target_arch",
    "snerf_kplanes_gather_fwd",
    "snerf_kplanes_gather_bwd",
    "snerf_spaced_bins",
    "snerf_weights_fwd",
    "snerf_weights_bwd",
    "snerf_pdf_resample",
    "snerf_mlp_param_count",
    "snerf_mlp_fwd",
    "snerf_mlp_bwd",
    "snerf_mlp_supported",
    "snerf_dense_fwd",
    "snerf_dense_bwd",
    "snerf_dense_bwd_fx",
    "snerf_dense_lp_supported",
    "snerf_dense_fwd_lp",
    "snerf_dense_bwd_lp",
    "snerf_render_fwd",
    "snerf_ray_train_fwd_bwd",
    "snerf_render_bwd",
    "snerf_distortion",
    "snerf_interlevel",
    "snerf_plane_reg",
    "snerf_adam_step",
    "snerf_adam_planes_step",
    "snerf_adam_planes_step_range",
    "snerf_render_mse_bwd",
    "snerf_tgrid_tv_fwd",
    "snerf_tgrid_tv_bwd",
    "snerf_tgrid_tv_fwd_bwd",
    "snerf_tgrid_tv_sign",
    "snerf_isg_maps",
    "snerf_adam_step_tv",
    "snerf_sample_pixels_uniform",
    "snerf_sort_rays_by_key",
    "snerf_kplanes_scatter_sorted_scales",
    "snerf_raygen",
    "snerf_aabb_collide",
    "snerf_tgrid_encode_fwd",
    "snerf_hashgrid_layout",
    "snerf_hashgrid_encode_fwd",
    "snerf_hashgrid_encode_bwd",
    "snerf_hashgrid_encode_bwd_fx",
    "snerf_tgrid_encode_bwd",
    "snerf_tgrid_encode_bwd_fx",
    "snerf_tgrid_encode_fwd_dydx",
    "snerf_tgrid_input_bwd",
    "snerf_tgrid_tile_plan_make",
    "snerf_tgrid_bwd_bin",
    "snerf_tgrid_bwd_tiles",
    "snerf_tgrid_bwd_tiles_adam",
    "snerf_tgrid_encode_bwd_levels",
    "snerf_hashgrid_tile_plan_make",
    "snerf_hashgrid_encode_bwd_levels",
    "snerf_hashgrid_bwd_bin",
    "snerf_hashgrid_bwd_tiles",
    "snerf_hashgrid_bwd_tiles_adam",
    "snerf_ist_maps",
    "snerf_ist_sample",
    "snerf_kplanes_sort_workspace",
    "snerf_kplanes_sort_samples",
    "snerf_kplanes_gradvec",
    "snerf_kplanes_scatter_sorted",
    "snerf_kplanes_gather_bwd_fx",
    "snerf_fx_to_float",
    "snerf_mlp_bwd_fx",
    "snerf_mlp_bwd_tile",
    "snerf_mlp_bwd_ws",
    "snerf_mlp_bwd_x16_quotient_ws",
    "snerf_mlp_gw_reduce",
    "snerf_mlp_gw_workspace_floats",
    "snerf_mlp_bwd_x16",
    "snerf_mlp_bwd_x16_quotient",
    "snerf_adam_prepare",
    "snerf_depth_loss",
    "snerf_urf_depth_loss",
    "snerf_kplanes_field_fwd",
    "snerf_kplanes_field_fwd_supported",
    "snerf_kplanes_density_fwd",
    "snerf_kplanes_density_fwd_supported",
    "snerf_kplanes_quotient_supported",
    "snerf_kplanes_quotient_prepare",
    "snerf_kplanes_scatter_quotient_scales",
    "snerf_kplanes_quotient_fixup",
    "snerf_nerfplayer_mix_fwd",
    "snerf_nerfplayer_mix_bwd",
    "snerf_nerfacto_head_input_fwd",
    "snerf_nerfacto_head_input_bwd",
    "snerf_trunc_exp_fwd",
    "snerf_trunc_exp_bwd",
    "snerf_basis_rgb_fwd",
    "snerf_basis_rgb_bwd",
    "snerf_comm_unique_id",
    "snerf_comm_create",
    "snerf_comm_destroy",
    "snerf_allreduce_grads",
]
