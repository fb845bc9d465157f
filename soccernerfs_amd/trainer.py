"""Fused K-Planes training step: the whole hot path as one fixed sequence of libsnerf launches.

What the reference does per step: Trainer.train_iteration -> VanillaPipeline.get_train_loss_dict ->
KPlanesModel.forward/get_loss_dict -> autograd backward -> 2x Adam -> scheduler
(NS/engine/trainer.py:383-412, NS/models/kplanes.py:349-452, NS/model_components/ray_samplers.py:559-600).
Here the same mathematics runs as ~30 hand-written HIP kernels on one stream with every buffer preallocated,
no autograd graph, no host synchronisation: forward, hand-derived backward (recompute instead of saving
activations), regulariser gradients, one flat gradient buffer (a single RCCL all-reduce when world_size > 1),
one fused Adam sweep that also clears the gradients.  `soccernerfs_amd.kplanes.KPlanesModel` is the
nerfstudio-shaped (autograd) face of the same kernels; tests check the two against each other and the oracle.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops
from .streams import side_stream
from .plane_set import PlaneSet
from .tcnn_compat import Network


@dataclass
class KPlanesTrainConfig:
    """Values of the `k-planes` preset (NS/configs/method_configs.py:481-560) unless overridden."""

    aabb_scale: float = 1.5
    spacetime_resolution: Sequence[int] = (64, 64, 64, 100)
    multiscale_res: Sequence[int] = (1, 2, 4, 8, 16)
    feature_dim: int = 32
    proposal_resolutions: Sequence[Sequence[int]] = ((128, 128, 128, 100), (256, 256, 256, 100))
    proposal_feature_dim: int = 8
    sigma_net_hidden_dim: int = 128
    rgb_net_hidden_dim: int = 64
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 128)
    num_nerf_samples_per_ray: int = 64
    use_single_jitter: bool = False
    near_plane: float = 0.0  # AABBBoxCollider default (kplanes.py:276-277)
    proposal_warmup: int = 5000
    proposal_update_every: int = 5
    proposal_weights_anneal_max_num_iters: int = 1000
    proposal_weights_anneal_slope: float = 10.0
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {
        "rgb_loss": 1.0, "interlevel_loss": 1.0, "distortion_loss": 0.001, "space_tv_loss": 0.0002,
        "time_smoothness_loss": 0.001, "sparse_transients_loss": 0.0001, "space_tv_proposal_loss": 0.0002,
        "time_smoothness_proposal_loss": 0.00001, "sparse_transients_proposal_loss": 0.0001, "depth_loss": 0.05})
    # depth supervision (NS/models/kplanes.py:162-172,395-409): active only when train_step / backward receive termination depths
    depth_sigma: float = 0.01
    is_euclidean_depth: bool = True
    # optimiser + schedule (method_configs.py:546-557)
    lr: float = 1e-2
    adam_eps: float = 1e-12
    warm_up_end: int = 512
    max_steps: int = 30000
    lr_alpha: float = 0.0
    seed: int = 0
    # MFMA operand type of every net (sigma_net, color_net, proposal sigma nets): "bf16" (default: what BASELINE config 2 names) / "fp16"
    # (tcnn's own) = 16-bit operands with fp32 accumulation (csrc/mlp_lp.hip, csrc/field_fused.hip); "fp32" = exact (the parity tests).
    # 30 k-step novel-view PSNR over 3 seeds: fp32 41.50 dB, bf16 41.49 dB (profiles/r02_psnr_ab.md).
    mlp_operands: str = "bf16"
    sigma_operands: Optional[str] = None   # per-net overrides of mlp_operands (A-B runs): field sigma_net / color_net / proposal nets
    color_operands: Optional[str] = None
    proposal_operands: Optional[str] = None
    # element type of the per-plane gradient vectors between the two passes of the sorted scatter: "fp32" (default) or "bf16" (half the
    # bytes of the step's largest intermediate, ~+1 % throughput; over 3 seeds it costs ~0.4 dB of novel-view PSNR with a +-0.8 dB
    # run-to-run spread against +-0.2 dB for fp32 vectors -- profiles/r02_psnr_ab.md -- so it stays opt-in)
    gvec_dtype: str = "fp32"
    # ---- execution switches (defaults = the measured best; bench.py / tools expose them for A-B runs) ----
    overlap: bool = True              # independent kernel chains on role streams (False: everything on the caller's stream)
    async_field_adam: bool = True     # field planes' optimiser sweep on its own stream under the NEXT step's proposal levels
    defer_prop: bool = True           # join the proposal chain only in front of the proposal planes' own optimiser kernels
    fused_ray_loss: bool = True       # train_step: the nerf level's weights / render / MSE / distortion / weights-backward as ONE launch
    #                                   (snerf_ray_train_fwd_bwd: bit-identical to the five kernels, ~0.08 ms less on the critical path)
    sorted_scatter: bool = True       # sorted / grouped plane-gradient scatter for the field (csrc/kplanes_sorted.hip)
    fuse_reg_into_adam: bool = True   # plane regularisers inside the optimiser sweep (ping-pong parameter buffers)
    shard_optimizer: bool = True      # world > 1: reduce-scatter -> Adam on a 1/world shard -> all-gather (False: one all-reduce)
    # weight gradients of the 16-bit MLP backward kernels through 16-replica workspaces (snerf_mlp_bwd_ws / snerf_mlp_gw_reduce): the flush of a
    # launch queues 16 same-address atomics instead of 256 (~25 us per launch), folded into the gradient buffer before the optimiser reads it
    mlp_grad_workspace: bool = True
    # proposal backward on updated steps: the two levels' kernels interleaved by stage (both net backwards before the two plane scatters) instead of
    # level by level.  Measured in round 5 (bench.py --interleave-prop-levels, same box, two runs each): 2.252 / 2.218 against 2.221 / 2.232 ms early
    # schedule -- no difference outside the run-to-run noise, although the traced step suggested 0.4 ms on updated steps: off
    interleave_proposal_levels: bool = False
    fix_capacity: Optional[int] = None  # quotient scatter: entries of the vanished-feature fix list (None: one per (sample, scale); ops.SortedScatter)
    exchange_chunks: int = 2          # world > 1, sharded: 2 = finest scale exchanged on its own, ahead of the rest (1: one exchange)
    grad_transport: str = "fp32"      # world > 1, sharded: "bf16" halves the reduce-scatter bytes (not the reference's fp32 DDP)
    param_transport: str = "fp32"     # world > 1, sharded: "bf16" gathers the parameter UPDATES in bf16
    # Single-GPU EMULATION of the two half-width transports' numerics (tools/train_psnr.py --emulate-transports; PSNR studies without a
    # multi-GPU node): "grad" rounds the field-plane gradient to bf16 before Adam (what a bf16 reduce-scatter hands the optimiser; one
    # rounding, where W ranks' ring sum makes ~log W of them), "param" applies new = old + bf16(new - old) to the field planes (exactly the bf16
    # update gather), "both".  Slow path: the plain sweep, no tile kernel.
    emulate_transports: str = ""
    # fixed-point (int64) gradient accumulation instead of float atomics: sums no longer depend on the order in which wavefronts
    # arrive, so two runs from one seed are bit-identical (debugging / reproducibility; ~2x slower steps)
    deterministic: bool = False
    # what a non-finite gradient does.  "skip_step": the whole optimiser step of that parameter group is skipped, as the
    # reference's GradScaler does (NS/engine/trainer.py:394-408, one found_inf per optimiser); "drop_elements": only the
    # non-finite elements are dropped (round-1 behaviour)
    nonfinite_policy: str = "skip_step"
    # gather -> sigma_net -> colour net as ONE kernel forward (csrc/field_fused.hip; 16-bit operands, the preset's net shapes): features and
    # activations stay on chip; for training it also writes the rounded feature tile (2 B / feature) and the 16 sigma_net outputs, which
    # is all the unfused backward kernels need.  Forward bit-identical to the unfused 16-bit kernels; 0.31 ms against 0.45 ms for
    # gather + sigma_net + color_net at the preset (profiles/r02_kernels.md).  Falls back to the unfused kernels when the shape / operand
    # type is outside what the fused kernel is built for (fp32 operands: the parity path).
    fused_field: bool = True
    # Quotient form of the field's sorted scatter (csrc/kplanes_sorted.hip, include/snerf.h): the gradient of plane q is (gfeat .* feat) ./ v_q
    # with v_q re-interpolated by pass B, so the second gather of all 30 planes and the 1 GB of per-plane gradient vectors (gradvec) go away:
    # ~1.6 GB less HBM traffic per step.  Equal to the product form to a few ulp (pass B recomputes the forward's v_q bit for bit; rows with an
    # exactly-zero feature take an exact fix-up).  Needs the sorted scatter, C = 32 and no deterministic mode; False = product form (A-B).
    quotient_scatter: bool = True
    # Round 4: G = gfeat .* feat is formed in the sigma_net backward's EPILOGUE from the 16-bit feature tile that kernel holds in LDS
    # (snerf_mlp_bwd_x16_quotient) instead of by a separate pass over fp32 copies of both tensors: no quotient_prepare launch on the critical
    # chain, and neither the forward's fp32 features nor gfeat cross HBM (~0.67 GB per step at the preset).  Needs the fused forward, the
    # quotient scatter and bf16 operands for sigma_net (fp16's narrow range would coarsen small features); G then carries the 2^-9 operand
    # rounding of the features.  False = round 3's flow (quotient_prepare on fp32 features; A-B).
    quotient_epilogue: bool = True
    # Round 4: each proposal level's density as ONE kernel (csrc/proposal_fused.hip: gather -> 8 -> 64 -> 1 net -> trunc_exp; bit-identical to
    # the two unfused kernels).  The [N,8] features go to HBM only on steps that update the proposal networks.  16-bit operands only.
    fused_proposal: bool = True
    # Round 4: inside train_step (single GPU) pass B of the field scatter is issued on the optimiser sweep's stream, in front of the sweep, instead of
    # on the caller's stream: the caller's stream is then free as soon as the sigma_net backward is queued, so the NEXT step's head (pixel draw, ray
    # generation, proposal levels) runs beside pass B (bound by float atomics) and has mostly finished when the sweep (bound by HBM) starts; and the
    # pass B -> sweep hand-over stays inside one stream.  False: pass B on the caller's stream (round 3; A-B).
    pass_b_beside_head: bool = True
    # Round 5 (A-B, off): the nerf level's sample sort needs the sample COORDINATES only, so it can be issued before the wait for the field planes'
    # optimiser sweep instead of after the field forward.  Measured: the colour-net backward it no longer runs beside drops from 0.17 to 0.065 ms in the
    # untraced step, but the sort under the HBM-bound sweep takes 0.3-0.7 ms and stretches the sweep from 0.82 to 0.97 ms -- steady state 2.03 / 2.07 ms
    # with it against 1.98 / 2.01 without (two runs each, one box; profiles/r05_step_offsets_early_sort.txt).
    sort_before_field_fwd: bool = False
    # Round 5 (A-B, off): pass B and the optimiser sweep of the field planes PIPELINED by scale (single GPU, inside train_step).  Pass B scatters one part
    # of the scales first; the sweep of those planes then starts on the idle "sort" stream while pass B goes on with the other part on the sweep's stream,
    # whose planes are swept behind it.  "coarse_first": scales 0..n-2, then the finest (~72 % of the floats at the preset); "fine_first": the finest scale
    # first, its big sweep beside the coarser scales' scatter.  Measured on one box (profiles/r05_bench_final.json vs r05_bench_pipelined.json): early
    # schedule 2.140 ms against 2.208 (-3 %), steady state 2.097 against 2.068 and trained state 1.813 against 1.817 (no gain where the proposal networks
    # update every 5th step); config 3 4.57 / 4.63 against 4.60 / 4.64.  Both kernels lean on the memory system, so what one gains the other loses: beside
    # the sweep pass B of the coarser scales stretches from 0.39 to 0.85 ms (fine_first), beside pass B the coarse sweep runs at 0.46 of the HBM peak instead
    # of 0.79 (coarse_first; profiles/r05_step_offsets_pipelined_*.txt).  Off: the gain is confined to the early schedule and inside the box-to-box spread.
    pipeline_sweep: str = ""


def anneal_value(step: int, max_iters: int, slope: float) -> float:
    """set_anneal callback (NS/models/kplanes.py:326-331)."""
    frac = min(max(step / max_iters, 0.0), 1.0)
    return (slope * frac) / ((slope - 1) * frac + 1)


def update_schedule(step: int, warmup: int, every: int) -> float:
    """NS/models/kplanes.py:254-259."""
    return min(max(every * min(max(step / warmup, 0.0), 1.0), 1.0), float(every))


def cosine_lr_factor(step: int, warm_up_end: int, max_steps: int, alpha: float) -> float:
    """CosineDecayScheduler (NS/engine/schedulers.py:126-141)."""
    if step < warm_up_end:
        return step / warm_up_end
    progress = (step - warm_up_end) / (max_steps - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class KPlanesTrainer:
    """Owns parameters (one flat fp32 buffer), Adam state and all work buffers for a fixed ray batch size R."""

    def __init__(self, cfg: KPlanesTrainConfig, num_rays: int, device="cuda:0", process_group=None):
        self.cfg, self.R, self.dev = cfg, num_rays, torch.device(device)
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        # world > 1: reduce-scatter + sharded Adam + all-gather for the field planes instead of one all-reduce (see dist.py)
        self.shard_optimizer = self.world > 1 and cfg.shard_optimizer
        # opt-in half-width transports of the sharded step: "bf16" rounds the field-plane gradient before the reduce-scatter /
        # gathers the parameter UPDATES in bf16; "fp32" (default) keeps the reference's DDP semantics
        self.grad_transport, self.param_transport = cfg.grad_transport, cfg.param_transport
        self._delta_pending = False
        # execution switches (KPlanesTrainConfig); plain attributes so that A-B tools can flip them between steps.
        # async_field_adam: the field planes' optimiser sweep runs on its own stream under the NEXT step's pixel draw / ray generation /
        # proposal levels (which read only the small segments); forward() joins it before the field gather, loss_dict() and
        # synchronize() join it for outside readers -- call synchronize() before reading parameters / Adam state / gradients from
        # outside a train step.  +3-5 % (bench.py --sync-adam for A-B).
        self.overlap, self.defer_prop = cfg.overlap, cfg.defer_prop
        self.async_field_adam = cfg.async_field_adam
        self._field_adam_done = None
        self._reg_in_adam = False
        self._render_deferred = False
        self._prop_pending = None
        self._depth = None
        self._exchange_started = False
        self.cabi_comm = None  # dist.CAbiComm: route the unsharded all-reduce through libsnerf's own RCCL communicator (bench.py --cabi-allreduce)
        self._fwd_fused = False
        self._grad_scale = 1.0
        self._reg_zeroed = False
        self._dyn_step = 0
        self._ar_work = self._reg_work = None
        self._side = {}  # role -> HIP stream, created on first use
        gen = torch.Generator().manual_seed(cfg.seed)
        a = cfg.aabb_scale
        self.aabb = [[-a, -a, -a], [a, a, a]]
        base = list(cfg.spacetime_resolution)
        reso = [[r * m for r in base[:3]] + base[3:] for m in cfg.multiscale_res]
        def mlp(din, dout, h, nh, act, operands):
            ncfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": h, "n_hidden_layers": nh}
            seed = int(torch.randint(0, 2**31, (1,), generator=gen))
            ops_ = operands or cfg.mlp_operands
            if operands is None and ops_ != "fp32":
                # the model-wide default applies where the 16-bit kernels are built for the shape; other shapes (e.g. the reference field's
                # own sigma_net_hidden_dim = 64 behind >= 2 scales) keep training with exact fp32 operands.  An explicit per-net
                # override (sigma / color / proposal_operands) is taken literally and raises in Network if unsupported.
                probe = _lib.MlpDesc()
                probe.d_in, probe.d_out, probe.hidden, probe.n_hidden = din, dout, h, nh
                probe.hidden_act, probe.out_act = 1, {"None": 0, "Sigmoid": 1}[act]
                probe.operands = {"bf16": 1, "fp16": 2}[ops_]
                if not _lib.lib().snerf_mlp_supported(C.byref(probe)):
                    import warnings

                    warnings.warn(f"{ops_} MFMA operands are not built for the {din} -> {h} x {nh} -> {dout} net: it runs with fp32 operands")
                    ops_ = "fp32"
            return Network(din, dout, ncfg, seed=seed, operands=ops_)

        self.field_planes = PlaneSet(cfg.feature_dim, reso, concat=True, a=0.1, b=0.5, generator=gen)
        self.sigma_net = mlp(cfg.feature_dim * len(reso), 16, cfg.sigma_net_hidden_dim, 1, "None", cfg.sigma_operands)
        self.color_net = mlp(15, 3, cfg.rgb_net_hidden_dim, 2, "Sigmoid", cfg.color_operands)
        self.prop_planes = [PlaneSet(cfg.proposal_feature_dim, [list(r)], concat=False, a=0.1, b=0.15, generator=gen)
                            for r in cfg.proposal_resolutions]
        self.prop_nets = [mlp(cfg.proposal_feature_dim, 1, 64, 1, "None", cfg.proposal_operands) for _ in cfg.proposal_resolutions]
        # ---- flatten: [proposal_networks | fields], each segment 16-B aligned ----
        self.segments = []  # (name, module, attr, offset, numel)
        off = 0
        for i, (pp, pn) in enumerate(zip(self.prop_planes, self.prop_nets)):
            for name, mod, attr in ((f"prop{i}.planes", pp, "planes"), (f"prop{i}.mlp", pn, "params")):
                n = getattr(mod, attr).numel()
                self.segments.append((name, mod, attr, off, n))
                off += _align4(n)
        self.n_proposal_params = off
        for name, mod, attr in (("field.planes", self.field_planes, "planes"), ("field.sigma", self.sigma_net, "params"),
                                ("field.color", self.color_net, "params")):
            n = getattr(mod, attr).numel()
            self.segments.append((name, mod, attr, off, n))
            if name == "field.planes":
                # padded so that the segment splits into `world` equal float4-aligned optimiser shards; the pad stays zero
                q = 4 * self.world
                self._field_seg = (off, n, (n + q - 1) // q * q)
                off += self._field_seg[2]
            else:
                off += _align4(n)
        self.n_params = off
        from .exchange_plan import kplanes_segment_sizes

        # the tensor-free restatement of this layout (what tests/test_exchange_plan_cpu.py checks at world 2 / 4 / 8) must describe THIS buffer
        sz = kplanes_segment_sizes(base, cfg.multiscale_res, cfg.feature_dim, cfg.proposal_resolutions, cfg.proposal_feature_dim,
                                   {"prop": self.prop_nets[0].params.numel(), "sigma": self.sigma_net.params.numel(), "color": self.color_net.params.numel()}, self.world)
        assert (sz["n_params"], sz["n_proposal_params"], sz["field_offset"], sz["field_floats"], sz["field_padded"]) == \
            (self.n_params, self.n_proposal_params) + tuple(self._field_seg), (sz, self.n_params, self._field_seg)
        self.params = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.views, self.gviews, self.mviews, self.vviews = {}, {}, {}, {}
        for name, mod, attr, o, n in self.segments:
            self.params[o:o + n].copy_(getattr(mod, attr).detach())
            self.gviews[name], self.mviews[name], self.vviews[name] = self.grads[o:o + n], self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n]
        self._repoint(self.params)
        # device-resident optimiser state per parameter group (= per torch optimiser of the reference, kplanes.py:311-316): skip-step flag,
        # Adam step counter, counters of skipped steps / dropped elements (include/snerf.h: snerf_adam_dyn)
        self._dyn = {"fields": ops.new_adam_dyn(self.dev), "proposal_networks": ops.new_adam_dyn(self.dev)}
        self._prepared = set()
        # deterministic mode: gradients accumulate as 64-bit fixed point (same layout as self.grads) and are converted once per step
        self.grads_fx = torch.zeros(off, dtype=torch.int64, device=self.dev) if cfg.deterministic else None
        # the regulariser-fused optimiser sweep reads neighbours of the OLD parameters: parameters ping-pong between two buffers
        self.fuse_reg_into_adam = cfg.fuse_reg_into_adam
        self._params_alt = torch.zeros_like(self.params)
        self._exchange = []  # per exchange chunk: dict(lo, hi, g_shard, p_shard, ...) -- see _plan_exchange
        # ---- work buffers ----
        R = num_rays
        S0, S1 = cfg.num_proposal_samples_per_ray
        S2 = cfg.num_nerf_samples_per_ray
        self.S = (S0, S1, S2)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        self.buf = {
            "sb": [f(R, s + 1) for s in self.S], "eb": [f(R, s + 1) for s in self.S],
            "dens": [f(R, s) for s in self.S], "w": [f(R, s) for s in self.S], "gw": [f(R, s) for s in self.S],
            "gdens": [f(R, s) for s in self.S],
            "pfeat": [f(R * S0, cfg.proposal_feature_dim), f(R * S1, cfg.proposal_feature_dim)],
            "pout": [f(R * S0, 1), f(R * S1, 1)],
            "gpfeat": [f(R * S0, cfg.proposal_feature_dim), f(R * S1, cfg.proposal_feature_dim)],
            "feat": f(R * S2, self.field_planes.out_dim), "gfeat": f(R * S2, self.field_planes.out_dim),
            "h": f(R * S2, 16), "gh": torch.zeros(R * S2, 16, dtype=torch.float32, device=self.dev),
            "rgb": f(R * S2, 3), "grgb": f(R * S2, 3),
            "rgb_out": f(R, 3), "acc": f(R), "depth": f(R), "g_rgb_out": f(R, 3),
            "dist_rays": f(R), "inter_rays": [f(R), f(R)], "sqerr": torch.zeros(R, dtype=torch.float32, device=self.dev),
            "depth_rays": [f(R), f(R), f(R)],
            "reg": torch.zeros(3, ops.REG_SLOTS, 16, dtype=torch.float32, device=self.dev),  # [field|prop0|prop1][slot][16]
        }
        self._timing, self._timing_all = None, False
        # sorted plane-gradient scatter for the main field (csrc/kplanes_sorted.hip): ~6x fewer atomic requests
        self.sorted_scatter = cfg.sorted_scatter
        self._gvec_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[cfg.gvec_dtype]
        self.quotient_scatter = bool(cfg.quotient_scatter and self.sorted_scatter and not cfg.deterministic
                                     and cfg.gvec_dtype == "fp32" and self.lib_quotient_ok(R * S2))
        # weight-gradient workspaces of the 16-bit MLP backward kernels (cfg.mlp_grad_workspace; not in deterministic mode: fixed-point cells
        # are order-independent already)
        self._mlp_nets = {"field.sigma": self.sigma_net, "field.color": self.color_net, **{f"prop{i}.mlp": n for i, n in enumerate(self.prop_nets)}}
        self._mlp_ws, self._ws_dirty = {}, set()
        if cfg.mlp_grad_workspace and not cfg.deterministic:
            for gname, net in self._mlp_nets.items():
                if net.desc.operands != 0:
                    self._mlp_ws[gname] = torch.zeros(int(_lib.lib().snerf_mlp_gw_workspace_floats(C.byref(net.desc))), dtype=torch.float32, device=self.dev)
        self._ss = ops.SortedScatter(self.field_planes, R * S2, self.dev, self._gvec_dtype, quotient=self.quotient_scatter,
                                     fix_capacity=cfg.fix_capacity)
        self._fix_peak_host = torch.zeros(1, dtype=torch.int32).pin_memory() if self.quotient_scatter else None
        self._ss.desc = self.field_planes.desc()
        if cfg.emulate_transports not in ("", "grad", "param", "both"):
            raise ValueError(f"emulate_transports must be '', 'grad', 'param' or 'both', got {cfg.emulate_transports!r}")
        self._sort_done = None
        self._passb_done = None
        self._passb_fine_done = None
        self.field_sweep_launches = 1  # launches of the field planes' sweep in the last optimiser step (2 when pipelined with pass B: cfg.pipeline_sweep)
        # (the proposal planes keep the sample-major scatter: they are small enough that their atomics are served by L2 -- 0.5 M of 19 M requests
        # reach memory -- while sorting 1.5 M samples x 6 planes cost ~0.6 ms: measured in round 1, the opt-in path was removed in round 3)
        self.step = 0                 # completed optimiser steps
        self._steps_since_update = 0  # ProposalNetworkSampler bookkeeping (ray_samplers.py:546-557)
        self.last = {}
        self.lib = _lib.lib()
        self._desc_field = self.field_planes.desc()
        self._desc_prop = [p.desc() for p in self.prop_planes]
        self.fused_field = bool(cfg.fused_field and self.lib.snerf_kplanes_field_fwd_supported(
            C.byref(self._desc_field), C.byref(self.sigma_net.desc), C.byref(self.color_net.desc)))
        if self.fused_field:  # the forward's operand-typed feature tile, kept for the unfused backward (snerf_mlp_bwd_x16)
            dt16 = torch.bfloat16 if self.sigma_net.desc.operands == 1 else torch.float16
            self.buf["feat16"] = torch.empty(R * self.S[2], self.field_planes.out_dim, dtype=dt16, device=self.dev)
        self.quotient_epilogue = bool(cfg.quotient_epilogue and self.quotient_scatter and self.fused_field and self.sigma_net.desc.operands == 1
                                      and self.sigma_net.desc.hidden == 128 and self.sigma_net.desc.n_hidden == 1)
        self._qg_step = False
        self.pass_b_beside_head = cfg.pass_b_beside_head
        self._in_train_step = False
        self.fused_proposal = bool(cfg.fused_proposal and all(self.lib.snerf_kplanes_density_fwd_supported(C.byref(dp), C.byref(net.desc))
                                                             for dp, net in zip(self._desc_prop, self.prop_nets)))
        self._keep_pfeat = True  # train_step clears it for steps that do not update the proposal networks
        if self.world > 1:
            self._plan_exchange()

    def _plan_exchange(self):
        """Sharded optimiser (world > 1): the field-plane segment is exchanged in CHUNKS so that the reduce-scatter of a chunk starts as soon
        as its scatter is complete and its shard-Adam + all-gather run while the next chunk is still on the links.  Chunk 0 = the finest
        scale (the tail of the segment: planes are laid out scale-major; ~72 % of the floats at the preset), scattered first; chunk 1 =
        everything in front of it.  Inside a chunk rank r owns floats [lo + r * len / world, lo + (r + 1) * len / world).
        cfg.exchange_chunks = 1 restores the single exchange (A-B)."""
        from .exchange_plan import exchange_chunks

        _, n, npad = self._field_seg
        f = lambda k, dt=torch.float32: torch.zeros(k, dtype=dt, device=self.dev)
        self._exchange = []
        # the arithmetic lives in exchange_plan.py (checked on the CPU at world 2 / 4 / 8 on the preset's sizes: tests/test_exchange_plan_cpu.py)
        for ch in exchange_chunks(npad, self._finest_offset(), self.world, self.cfg.exchange_chunks, len(self.cfg.multiscale_res)):
            shard = ch["shard"]
            self._exchange.append({"lo": ch["lo"], "hi": ch["hi"], "shard": shard, "g_shard": f(shard), "p_shard": f(shard), "rs": None, "ag": None,
                                   "g16": None, "g16_shard": None, "d16_full": None, "d16_shard": None})

    def _repoint(self, flat: torch.Tensor):
        """Make `flat` the live parameter buffer: module parameters and self.views alias its segments."""
        self.params = flat
        for name, mod, attr, o, n in self.segments:
            getattr(mod, attr).data = flat[o:o + n]
            self.views[name] = flat[o:o + n]

    # -------------------------------------------------------------------------------------------
    def enable_kernel_timing(self, names=None):
        """Record HIP events (on the launch stream) around kernel groups; `names` = None times every group.
        Read back with `kernel_times_ms()` (synchronises)."""
        self._timing = {} if names is None else {n: [] for n in names}
        self._timing_all = names is None

    def disable_kernel_timing(self):
        self._timing = None

    def kernel_times_ms(self) -> Dict[str, Tuple[float, int]]:
        """name -> (mean milliseconds per launch, launches)."""
        torch.cuda.synchronize()
        out = {}
        for k, evs in (self._timing or {}).items():
            if evs:
                out[k] = (sum(a.elapsed_time(b) for a, b in evs) / len(evs), len(evs))
        return out

    class _Span:
        def __init__(self, tr, name):
            self.tr, self.name = tr, name

        def __enter__(self):
            t = self.tr._timing
            self.on = t is not None and (self.tr._timing_all or self.name in t)
            if self.on:
                self.a = torch.cuda.Event(enable_timing=True)
                self.a.record()

        def __exit__(self, *exc):
            if self.on:
                b = torch.cuda.Event(enable_timing=True)
                b.record()
                self.tr._timing.setdefault(self.name, []).append((self.a, b))

    def _span(self, name):
        return KPlanesTrainer._Span(self, name)

    COMM_WAIT_SPANS = ("comm_wait.reduce_scatter", "comm_wait.all_gather", "comm_wait.all_reduce", "comm_wait.flags", "allreduce_grads")

    def _comm_wait(self, work, name: str):
        """The current stream waits for an asynchronous collective.  Under kernel timing the wait sits between two events on that
        stream: the first fires when the chain has nothing left to run, the second when the collective is complete, so their distance
        is the EXPOSED communication time of this wait (0 when the collective finished under the kernels before it)."""
        with self._span(name):
            work.wait()

    def link_bytes_per_step(self) -> Dict[str, float]:
        """Bytes this rank SENDS over the links per optimiser step (= bytes it receives), by collective, from the segment sizes:
        reduce-scatter and all-gather of n elements move (W-1)/W * n * elt each, an all-reduce twice that (ring or direct: the same
        per-rank volume).  World 1: all zero."""
        from .exchange_plan import link_bytes

        return link_bytes(self.world, self.n_params, self._field_seg[2], self.buf["reg"][0].numel(), self._sharded(), self.grad_transport, self.param_transport)

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def _gather(self, desc, planes, coords, N, out):
      with self._span("kplanes_gather_fwd.field" if desc is self._desc_field else "kplanes_gather_fwd.prop"):
        _lib.check(self.lib.snerf_kplanes_gather_fwd(C.byref(desc), self._p(planes), C.byref(coords), C.c_int64(N), self._p(out), self._st), "gather_fwd")

    def _fx(self, gview: torch.Tensor) -> torch.Tensor:
        """The fixed-point cells behind a view of self.grads (deterministic mode)."""
        o = gview.storage_offset() - self.grads.storage_offset()
        return self.grads_fx[o:o + gview.numel()]

    def _scatter(self, desc, planes, coords, N, gout, gplanes):
      with self._span("kplanes_gather_bwd.field" if desc is self._desc_field else "kplanes_gather_bwd.prop"):
        if self.grads_fx is not None:
            _lib.check(self.lib.snerf_kplanes_gather_bwd_fx(C.byref(desc), self._p(planes), C.byref(coords), C.c_int64(N), self._p(gout),
                                                            self._p(self._fx(gplanes)), self._st), "gather_bwd_fx")
        else:
            _lib.check(self.lib.snerf_kplanes_gather_bwd(C.byref(desc), self._p(planes), C.byref(coords), C.c_int64(N), self._p(gout), self._p(gplanes),
                                                         self._st), "gather_bwd")

    def _mlp_fwd(self, net, X, ldx, N, Y, ldy, aux_col=-1, aux=None):
      with self._span(f"mlp_fwd.{net.desc.d_in}x{net.desc.hidden}x{net.desc.n_hidden}"):
        _lib.check(self.lib.snerf_mlp_fwd(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(Y), ldy, aux_col,
                                          self._p(aux) if aux is not None else None, self._st), "mlp_fwd")

    def _mlp_bwd(self, net, gname, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx, x16=False):
      with self._span(f"mlp_bwd.{net.desc.d_in}x{net.desc.hidden}x{net.desc.n_hidden}"):
        ws = self._mlp_ws.get(gname) if not x16 else None
        if ws is not None:
            self._ws_dirty.add(gname)
            _lib.check(self.lib.snerf_mlp_bwd_ws(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N),
                                                 self._p(gY) if gY is not None else None, ldgy, aux_col, self._p(gaux) if gaux is not None else None,
                                                 self._p(gX) if gX is not None else None, ldgx, self._p(ws), self._st), "mlp_bwd_ws")
            return
        fn = self.lib.snerf_mlp_bwd_fx if self.grads_fx is not None else (self.lib.snerf_mlp_bwd_x16 if x16 else self.lib.snerf_mlp_bwd)
        if x16 and self.grads_fx is not None:  # deterministic mode: the fixed-point kernel takes fp32 inputs (exact image of the 16-bit tile)
            X = X.float()
        gw = self._fx(self.gviews[gname]) if self.grads_fx is not None else self.gviews[gname]
        _lib.check(fn(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N),
                      self._p(gY) if gY is not None else None, ldgy, aux_col, self._p(gaux) if gaux is not None else None,
                      self._p(gX) if gX is not None else None, ldgx, self._p(gw), self._st), "mlp_bwd")

    def _reduce_mlp_grads(self, names):
        """Folds the weight-gradient workspaces of `names` into self.grads (current stream) and clears them; only those a backward kernel of this
        step has written."""
        for gname in names:
            if gname in self._ws_dirty:
                net = self._mlp_nets[gname]
                with self._span("mlp_gw_reduce"):
                    _lib.check(self.lib.snerf_mlp_gw_reduce(C.byref(net.desc), self._p(self._mlp_ws[gname]), self._p(self.gviews[gname]), self._st), "mlp_gw_reduce")
                self._ws_dirty.discard(gname)

    def _resample(self, lvl, rand, anneal):
        """density[lvl] -> weights[lvl] (stored) -> PDF sample level lvl+1 bins."""
        b, a = self.buf, _lib.ResampleArgs()
        a.density, a.ebins_prev, a.weights_out = b["dens"][lvl].data_ptr(), b["eb"][lvl].data_ptr(), b["w"][lvl].data_ptr()
        a.sbins_prev, a.nears, a.fars = b["sb"][lvl].data_ptr(), self.rays["nears"].data_ptr(), self.rays["fars"].data_ptr()
        if rand is None:
            a.u_mode = 2
        else:
            a.u_mode, a.u_or_rand, a.rand_cols = 1, rand.data_ptr(), rand.shape[-1]
        a.sbins_out, a.ebins_out = b["sb"][lvl + 1].data_ptr(), b["eb"][lvl + 1].data_ptr()
        a.R, a.S_prev, a.S, a.kind = self._fwd_rays, self.S[lvl], self.S[lvl + 1], 0
        a.anneal, a.histogram_padding, a.eps = anneal, 0.01, 1e-5
        with self._span("pdf_resample"):
            _lib.check(self.lib.snerf_pdf_resample(C.byref(a), self._st), "pdf_resample")

    # -------------------------------------------------------------------------------------------
    def forward(self, rays: Dict[str, torch.Tensor], rng: Optional[Dict[str, torch.Tensor]], anneal: float, training: bool = True,
                defer_render: bool = False):
        """rays: origins [R,3], directions [R,3], times [R,1] (+ nears/fars, else the AABB collider runs).
        rng (training): t_rand [R,S0+1]|[R,1], u (list of 2 draws [R,S+1]|[R,1]), bg [R,3].
        defer_render (train_step only): stop after the field; backward() then runs weights + compositing + the ray losses' backward in one
        launch and the returned rgb buffer is filled by it."""
        cfg, b = self.cfg, self.buf
        R = rays["origins"].shape[0]  # <= self.R: the work buffers are row-major, their first R rows are used (eval chunks)
        assert R <= self.R and (training is False or R == self.R), "training batches must have exactly the configured number of rays"
        self._fwd_rays = R
        self._st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # the kernels read these buffers through raw pointers: insist on contiguous fp32 device tensors of the expected shapes
        rays = dict(rays)
        for k, cols in (("origins", 3), ("directions", 3), ("times", 1)):
            rays[k] = ops._f32c(rays[k], f"rays[{k!r}]")
            if rays[k].numel() != R * cols:
                raise RuntimeError(f"rays[{k!r}] must hold {R} x {cols} values, got {tuple(rays[k].shape)}")
        if training:
            rng = dict(rng)
            rng["t_rand"], rng["bg"] = ops._f32c(rng["t_rand"], "rng['t_rand']"), ops._f32c(rng["bg"], "rng['bg']")
            rng["u"] = [ops._f32c(u, "rng['u']") for u in rng["u"]]
            if rng["bg"].numel() != 3 * R or any(u.shape[0] != R for u in rng["u"]) or rng["t_rand"].shape[0] != R:
                raise RuntimeError("rng draws must have one row per ray")
        o, d, t = rays["origins"], rays["directions"], rays["times"].reshape(-1)
        if "nears" not in rays:
            rays["nears"], rays["fars"] = ops.aabb_collide(o, d, self.aabb, cfg.near_plane, training)
        else:
            rays["nears"], rays["fars"] = ops._f32c(rays["nears"], "rays['nears']"), ops._f32c(rays["fars"], "rays['fars']")
        self.rays = rays
        t_rand = rng["t_rand"] if training else None
        _lib.check(self.lib.snerf_spaced_bins(self._p(rays["nears"]), self._p(rays["fars"]), self._p(t_rand) if t_rand is not None else None,
                                              t_rand.shape[-1] if t_rand is not None else 0, R, self.S[0], 0, self._p(b["sb"][0]), self._p(b["eb"][0]),
                                              self._st), "spaced_bins")
        self._coords = []
        for lvl in range(3):
            rescale = lvl == 2  # proposal fields keep [0,1] coordinates (kplanes_field.py:440), the main field maps to [-1,1] (:283-284)
            co = ops.coords_from_rays(o, d, t, b["eb"][lvl], self.aabb, rescale)
            self._coords.append(co)
            N = R * self.S[lvl]
            if lvl < 2 and self.fused_proposal:
                net = self.prop_nets[lvl]
                with self._span("kplanes_density_fwd"):
                    _lib.check(self.lib.snerf_kplanes_density_fwd(C.byref(self._desc_prop[lvl]), self._p(self.prop_planes[lvl].planes), C.byref(co), C.c_int64(N),
                                                                  C.byref(net.desc), self._p(net.params), self._p(b["dens"][lvl]),
                                                                  self._p(b["pfeat"][lvl]) if training and self._keep_pfeat else None, self._st),
                               "kplanes_density_fwd")
                self._resample(lvl, rng["u"][lvl] if training else None, anneal)
            elif lvl < 2:
                self._gather(self._desc_prop[lvl], self.prop_planes[lvl].planes, co, N, b["pfeat"][lvl])
                self._mlp_fwd(self.prop_nets[lvl], b["pfeat"][lvl], cfg.proposal_feature_dim, N, b["pout"][lvl], 1, 0, b["dens"][lvl])
                self._resample(lvl, rng["u"][lvl] if training else None, anneal)
            else:
                will_sort = bool(training and self.sorted_scatter and self.grads_fx is None and R == self.R)
                sort_early = bool(will_sort and self.cfg.sort_before_field_fwd and self._field_adam_done is not None)
                if sort_early:
                    self._issue_sort(co)
                self._wait_params()
                # a training step's fused forward leaves the feature tile (16-bit), the sigma_net outputs and (quotient scatter) the fp32 features
                # behind for the unfused backward kernels
                self._fwd_fused = self.fused_field
                keep = training
                # G comes out of the sigma_net backward's epilogue (quotient_epilogue): the fp32 features are then not needed at all
                self._qg_step = bool(self._fwd_fused and self.quotient_epilogue and will_sort)
                if self._fwd_fused:
                    with self._span("kplanes_field_fwd"):
                        _lib.check(self.lib.snerf_kplanes_field_fwd(C.byref(self._desc_field), self._p(self.field_planes.planes), C.byref(co), C.c_int64(N),
                                                                    C.byref(self.sigma_net.desc), self._p(self.sigma_net.params), C.byref(self.color_net.desc),
                                                                    self._p(self.color_net.params), self._p(b["dens"][2]), self._p(b["rgb"]),
                                                                    self._p(b["feat16"]) if keep else None, self._p(b["h"]) if keep else None,
                                                                    self._p(b["feat"]) if keep and self.quotient_scatter and not self._qg_step else None, self._st),
                                   "kplanes_field_fwd")
                else:
                    self._gather(self._desc_field, self.field_planes.planes, co, N, b["feat"])
                if will_sort and not sort_early:
                    # Started AFTER the (memory-bound) gather: it runs under the MLP backward kernels that follow on the main stream.
                    self._issue_sort(co)
                if not self._fwd_fused:
                    self._mlp_fwd(self.sigma_net, b["feat"], self.field_planes.out_dim, N, b["h"], 16, 15, b["dens"][2])
                    self._mlp_fwd(self.color_net, b["h"], 16, N, b["rgb"], 3)
                self._render_deferred = bool(training and defer_render)
                if not self._render_deferred:
                    _lib.check(self.lib.snerf_weights_fwd(self._p(b["dens"][2]), self._p(b["eb"][2]), R, self.S[2], self._p(b["w"][2]), self._st), "weights_fwd")
        if self._render_deferred:
            return b["rgb_out"][:R]
        a = _lib.RenderArgs()
        a.weights, a.rgb, a.ebins = b["w"][2].data_ptr(), b["rgb"].data_ptr(), b["eb"][2].data_ptr()
        if training:
            a.bg_mode, a.bg = 0, rng["bg"].data_ptr()
        else:
            a.bg_mode = 1
        a.R, a.S, a.training = R, self.S[2], int(training)
        a.rgb_out, a.acc_out, a.depth_median = b["rgb_out"].data_ptr(), b["acc"].data_ptr(), b["depth"].data_ptr()
        _lib.check(self.lib.snerf_render_fwd(C.byref(a), self._st), "render_fwd")
        return b["rgb_out"][:R]

    def _issue_sort(self, co):
        """Sort the nerf-level samples per (scale, plane) on the "sort" stream, behind what the caller's stream holds (the sample coordinates) and
        behind the last reader of the sorted records on another stream (pass B of the previous step, when it ran on the sweep's stream)."""
        main = torch.cuda.current_stream()
        st = self._stream("sort")
        st.wait_stream(main)
        if self._passb_done is not None:
            st.wait_event(self._passb_done)
            self._passb_done = None
        with KPlanesTrainer._On(self, st), self._span("kplanes_sort"):
            self._ss.sort(co, self._st)
        self._sort_done = st.record_event()

    # ---- stream helpers: kernels bound by different units overlap on separate HIP streams ----
    class _On:
        """Run the enclosed launches on `stream` (torch's current stream AND the stream handed to libsnerf)."""

        def __init__(self, tr, stream):
            self.tr, self.stream = tr, stream

        def __enter__(self):
            self.prev = self.tr._st
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
            self.tr._st = C.c_void_p(self.stream.cuda_stream)

        def __exit__(self, *exc):
            self.tr._st = self.prev
            self.ctx.__exit__(*exc)

    def _stream(self, role: str):
        """Side stream by role, created on first use (so the common path holds main + "sort" + "prop" + "adam" only: HIP multiplexes the
        normal-priority streams onto 3 hardware queues beside the null stream's -- tools/debug_queues.py -- and chains sharing a queue do
        not overlap; a high-priority stream would get a fifth queue, which made every step slower -- profiles/r02_kernels.md section 8)."""
        if role not in self._side:
            self._side[role] = side_stream(self.dev, role)  # one stream per role for the whole PROCESS (streams.py: hardware queues are few)
        return self._side[role]

    def _reg_sweep(self):
        """Plane regularisers: values + gradients in one sweep per plane set (kplanes.py:430-446).  The flat gradient buffer
        is still zero when this runs (Adam cleared it), so the sweep STORES its gradient and the scatters add on top."""
        b, co = self.buf, self.cfg.loss_coefficients
        b["reg"].zero_()
        with self._span("plane_reg.all"):
            _lib.check(self.lib.snerf_plane_reg(C.byref(self._desc_field), self._p(self.field_planes.planes), self._p(self.gviews["field.planes"]),
                                                co["space_tv_loss"], co["time_smoothness_loss"], co["sparse_transients_loss"], self._p(b["reg"][0]),
                                                ops.REG_SLOTS, 1, self._st), "plane_reg")
            for lvl in range(2):
                _lib.check(self.lib.snerf_plane_reg(C.byref(self._desc_prop[lvl]), self._p(self.prop_planes[lvl].planes),
                                                    self._p(self.gviews[f"prop{lvl}.planes"]), co["space_tv_proposal_loss"],
                                                    co["time_smoothness_proposal_loss"], co["sparse_transients_proposal_loss"],
                                                    self._p(b["reg"][1 + lvl]), ops.REG_SLOTS, 1, self._st), "plane_reg")

    def lib_quotient_ok(self, N: int) -> bool:
        d = self.field_planes.desc()
        return bool(_lib.lib().snerf_kplanes_quotient_supported(C.byref(d), C.c_int64(N)))

    def _scatter_field_scales(self, co, lo: int, hi: int, fixup: bool = True):
        """Pass B of the field's sorted scatter for scales [lo, hi): product form (gradient vectors from gradvec / the fused backward) or
        quotient form (G + the exact terms of the listed vanished-feature elements; fixup = False: the caller has issued the fix-up itself)."""
        ss, b = self._ss, self.buf
        with self._span("kplanes_scatter_sorted.field"):
            if self.quotient_scatter:
                if fixup:
                    ss.quotient_fixup_scales(self.field_planes.planes, co, self.gviews["field.planes"], lo, hi, self._st)
                ss.quotient_pass_b_scales(self.field_planes.planes, self.gviews["field.planes"], lo, hi, self._st)
            else:
                _lib.check(self.lib.snerf_kplanes_scatter_sorted_scales(C.byref(ss.desc), C.c_int64(ss.N), self._p(ss.gvec), ss.gvec_bf16, self._p(ss.sorted_rec),
                                                                        self._p(self.gviews["field.planes"]), lo, hi, self._st), "scatter_sorted")

    def _field_backward_chunk(self, r0: int, r1: int):
        """colour-net bwd -> sigma-net bwd -> plane scatter for rays [r0, r1) of the nerf level."""
        b, S2, F = self.buf, self.S[2], self.field_planes.out_dim
        n0, N = r0 * S2, (r1 - r0) * S2
        sl = lambda t: t[n0:n0 + N]
        # colour net: X = h[:, :15] (stride 16); its gX lands in gh[:, :15]; gh[:, 15] stays 0 (density enters through gaux)
        self._mlp_bwd(self.color_net, "field.color", sl(b["h"]), 16, N, sl(b["grgb"]), 3, -1, None, sl(b["gh"]), 16)
        qg = bool(self._qg_step and self._sort_done is not None and r0 == 0 and r1 == self.R)
        if qg:
            ss = self._ss
            k = ss.next_fix_counter()
            with self._span(f"mlp_bwd.{self.sigma_net.desc.d_in}x{self.sigma_net.desc.hidden}x{self.sigma_net.desc.n_hidden}"):
                ws = self._mlp_ws.get("field.sigma")
                if ws is not None:
                    self._ws_dirty.add("field.sigma")
                fnq = self.lib.snerf_mlp_bwd_x16_quotient_ws if ws is not None else self.lib.snerf_mlp_bwd_x16_quotient
                _lib.check(fnq(C.byref(self.sigma_net.desc), self._p(self.sigma_net.params), self._p(b["feat16"]), F, C.c_int64(N),
                               self._p(b["gh"]), 16, 15, self._p(b["gdens"][2]), self._p(ss.G), F, self._p(ss.fix_list),
                               ss.fix_capacity, self._p(ss.fix_count), self._p(ss.fix_counts[1 - k:2 - k]),
                               self._p(ws if ws is not None else self.gviews["field.sigma"]), self._st), "mlp_bwd_x16_quotient")
        else:
            self._mlp_bwd(self.sigma_net, "field.sigma", sl(b["feat16"] if self._fwd_fused else b["feat"]), F, N, sl(b["gh"]), 16, 15, b["gdens"][2][r0:r1],
                          sl(b["gfeat"]), F, x16=self._fwd_fused)
        rays = self.rays
        co = ops.coords_from_rays(rays["origins"][r0:r1], rays["directions"][r0:r1], rays["times"].reshape(-1)[r0:r1], b["eb"][2][r0:r1], self.aabb, True)
        if self.sorted_scatter and self.grads_fx is None and self._sort_done is not None and r0 == 0 and r1 == self.R:
            torch.cuda.current_stream().wait_event(self._sort_done)
            ss = self._ss
            ns = len(self.cfg.multiscale_res)
            beside = bool(self.pass_b_beside_head and self._in_train_step and self.world == 1 and self.overlap and self.async_field_adam and self._reg_in_adam
                          and not self.cfg.emulate_transports and not (self._sharded() and len(self._exchange) == 2))
            passb_issued = False
            if self.quotient_scatter:
                if not qg:
                    with self._span("kplanes_quotient_prepare"):
                        ss.quotient_prepare(b["gfeat"], b["feat"], self._st)
                if beside:
                    # (r05) pass B goes FIRST, on the sweep's stream, behind nothing but what it reads: G (the sigma_net backward, just queued) and the
                    # sorted records.  The sweep's forerunners and the fix-up below used to sit between the sigma_net backward and pass B on the critical
                    # chain (~35 us of tiny kernels and launch gaps per step, profiles/r05_timeline_step.txt); they only have to precede the SWEEP, which
                    # waits for everything the caller's stream holds when it is launched (_adam_field_range).
                    main = torch.cuda.current_stream()
                    st = self._stream("adam")
                    st.wait_event(main.record_event())
                    st.wait_event(self._sort_done)
                    with KPlanesTrainer._On(self, st):
                        if self.cfg.pipeline_sweep and ns > 1 and not self.cfg.emulate_transports:
                            fine_first = self.cfg.pipeline_sweep == "fine_first"
                            first, second = ((ns - 1, ns), (0, ns - 1)) if fine_first else ((0, ns - 1), (ns - 1, ns))
                            self._scatter_field_scales(co, *first, fixup=False)  # the sweep of these scales' planes starts when this is done
                            self._passb_fine_done = (st.record_event(), fine_first)
                            self._scatter_field_scales(co, *second, fixup=False)
                        else:
                            self._scatter_field_scales(co, 0, ns, fixup=False)
                    self._passb_done = st.record_event()
                    passb_issued = True
                if self.world == 1 and self._reg_in_adam:
                    # the optimiser sweep's two tiny forerunners (skip decision of the group, zeroed regulariser slots) depend on nothing pass B
                    # produces: issued here they are off the scatter -> sweep hand-over
                    cfgl = self.cfg
                    self._prepare_group("fields", cfgl.lr * cosine_lr_factor(self.step, cfgl.warm_up_end, cfgl.max_steps, cfgl.lr_alpha))
                    if not self._reg_zeroed:
                        b["reg"].zero_()
                        self._reg_zeroed = True
            else:
              with self._span("kplanes_gradvec.field"):
                _lib.check(self.lib.snerf_kplanes_gradvec(C.byref(ss.desc), self._p(self.field_planes.planes), C.byref(co), C.c_int64(ss.N),
                                                          self._p(b["gfeat"]), self._p(ss.gvec), ss.gvec_bf16, self._st), "gradvec")
            if self._sharded() and len(self._exchange) == 2:
                # finest scale first: its reduce-scatter (chunk 0) is on the links while the coarser scales are still being scattered
                self._scatter_field_scales(co, ns - 1, ns)
                self._start_field_grad_exchange(0)
                self._scatter_field_scales(co, 0, ns - 1)
                self._start_field_grad_exchange(1)
                self._exchange_started = True
                return
            if beside:
                # only inside train_step: the one consumer of these gradients is then the sweep, queued behind pass B on the same stream (a
                # caller of backward() may read the gradient buffer from ITS stream)
                # The fix-up reads the sample coordinates (the caller's ray tensors, the nerf level's bin edges, which the next step's head overwrites):
                # it stays on the caller's stream; like pass B it only ADDS to the gradient planes, so the two may run side by side.
                if self.quotient_scatter:
                    self._ss.quotient_fixup_scales(self.field_planes.planes, co, self.gviews["field.planes"], 0, ns, self._st)
                if not passb_issued:  # product form: pass B reads the gradient vectors gradvec has just written
                    main = torch.cuda.current_stream()
                    st = self._stream("adam")
                    st.wait_stream(main)
                    with KPlanesTrainer._On(self, st):
                        self._scatter_field_scales(co, 0, ns, fixup=False)
                    self._passb_done = st.record_event()
            else:
                self._scatter_field_scales(co, 0, ns)
        else:
            self._scatter(self._desc_field, self.field_planes.planes, co, N, sl(b["gfeat"]), self.gviews["field.planes"])

    def _depth_loss(self, lvl: int, with_grad: bool):
        """ds_nerf depth loss of one sampling level (kplanes.py:395-409: every level, weight 1/3); its gradient is ADDED to gw[lvl]."""
        if self._depth is None:
            return
        cfg, b, R = self.cfg, self.buf, self.R
        dn = None if cfg.is_euclidean_depth else self.rays["directions_norm"]
        _lib.check(self.lib.snerf_depth_loss(self._p(b["w"][lvl]), self._p(b["eb"][lvl]), self._p(self._depth), self._p(dn) if dn is not None else None,
                                             cfg.depth_sigma, R, self.S[lvl], cfg.loss_coefficients["depth_loss"] / (3 * R), self._p(b["depth_rays"][lvl]),
                                             self._p(b["gw"][lvl]) if with_grad else None, 1, self._st), "depth_loss")

    def _proposal_backward(self, proposal_grads: bool):
        """Proposal supervision (interlevel loss); gradients only on `updated` steps (ray_samplers.py:573,587-592).  The two levels touch disjoint
        buffers and parameter segments, so their kernels may be issued level by level (default) or interleaved by stage
        (cfg.interleave_proposal_levels: both weights backwards, both net backwards, then the two plane scatters).  In the TRACED step
        (profiles/r05_timeline_step.txt) level 1's net backward -- a 512-thread, 90-KB workgroup per CU -- lands under the field planes' optimiser
        sweep, whose small workgroups leave it no room (0.65 ms for a 0.03-ms kernel), and the chain ends after the sweep; the untraced step did
        not get faster with the interleaved order (see the config field), so the order stays as it was."""
        cfg, b, R, co = self.cfg, self.buf, self.R, self.cfg.loss_coefficients
        S2 = self.S[2]
        for lvl in (0, 1):
            Sp = self.S[lvl]
            _lib.check(self.lib.snerf_interlevel(self._p(b["sb"][2]), self._p(b["w"][2]), S2, self._p(b["sb"][lvl]), self._p(b["w"][lvl]), Sp, R,
                                                 co["interlevel_loss"] / (R * S2), self._p(b["inter_rays"][lvl]),
                                                 self._p(b["gw"][lvl]) if proposal_grads else None, self._st), "interlevel")
            self._depth_loss(lvl, with_grad=proposal_grads)  # adds to the interlevel gradient just written
        if not proposal_grads:
            return
        wb = lambda lvl: _lib.check(self.lib.snerf_weights_bwd(self._p(b["dens"][lvl]), self._p(b["eb"][lvl]), self._p(b["gw"][lvl]), R, self.S[lvl],
                                                               self._p(b["gdens"][lvl]), 0, self._p(self._dyn["proposal_networks"]), self._st), "weights_bwd")
        nb = lambda lvl: self._mlp_bwd(self.prop_nets[lvl], f"prop{lvl}.mlp", b["pfeat"][lvl], cfg.proposal_feature_dim, R * self.S[lvl], None, 1, 0,
                                       b["gdens"][lvl], b["gpfeat"][lvl], cfg.proposal_feature_dim)
        sc = lambda lvl: self._scatter(self._desc_prop[lvl], self.prop_planes[lvl].planes, self._coords[lvl], R * self.S[lvl], b["gpfeat"][lvl],
                                       self.gviews[f"prop{lvl}.planes"])
        if cfg.interleave_proposal_levels:
            for stage in (wb, nb, sc):
                for lvl in (0, 1):
                    stage(lvl)
        else:
            for lvl in (0, 1):
                for stage in (wb, nb, sc):
                    stage(lvl)

    def backward(self, target: torch.Tensor, rng: Dict[str, torch.Tensor], proposal_grads: bool, include_reg: bool = True,
                 defer_prop_join: bool = False, depth: Optional[torch.Tensor] = None):
        """Accumulates d(total loss)/d(params) into self.grads (which must be zero on entry: Adam clears it); fills
        self.last with the (scaled) loss terms.

        With `self.overlap` (default) independent kernel chains run on side streams so that kernels bound by different units
        overlap: the regulariser sweep (HBM stream), the proposal-level backward (MFMA + atomics) and the field backward split
        into ray chunks whose MLP backward (MFMA) runs under the previous chunk's plane scatter (memory-side atomics)."""
        cfg, b, R, co = self.cfg, self.buf, self.R, self.cfg.loss_coefficients
        S2 = self.S[2]
        main = torch.cuda.current_stream()
        self._reg_in_adam = not include_reg  # train_step: the regularisers' values and gradients come out of the optimiser sweep
        # depth supervision: termination depths [R] (batch["depth_image"]); None or a zero coefficient switches the term off
        self._depth = ops._f32c(depth, "depth").reshape(-1) if depth is not None and co.get("depth_loss", 0) > 0 else None
        overlap = self.overlap
        sharded = self._sharded()  # the field-plane gradient leaves for the reduce-scatter as soon as it is complete, and the
        #                            proposal backward runs AFTER it, under the collective
        joins = []
        reg_done = None
        if include_reg:
            if overlap:
                st = self._stream("reg")
                st.wait_stream(main)
                with KPlanesTrainer._On(self, st):
                    self._reg_sweep()
                reg_done = st.record_event()  # every scatter adds on top of the STORED regulariser gradient
                joins.append(st)
            else:
                self._reg_sweep()

        def proposal_chain(after=None):
            st = self._stream("prop")
            st.wait_stream(main) if after is None else st.wait_event(after)
            if reg_done is not None:
                st.wait_event(reg_done)
            # (the two levels' chains are independent, but side by side on two streams they were no faster: 2.78 vs 2.72 ms; on a
            # high-priority stream -- a fifth hardware queue -- the whole step fell to 3.69 ms: profiles/r02_kernels.md section 8)
            with KPlanesTrainer._On(self, st):
                self._proposal_backward(proposal_grads)
            joins.append(st)

        target = ops._f32c(target, "target")
        if target.numel() != 3 * R:
            raise RuntimeError(f"target must be [{R}, 3], got {tuple(target.shape)}")
        rng = dict(rng, bg=ops._f32c(rng["bg"], "rng['bg']"))
        self.last = {}
        if self._render_deferred:
            # forward() stopped after the field (train_step): weights -> compositing -> MSE / distortion backward -> weights backward, one launch.
            # The proposal chain reads the nerf level's weights, so it starts behind this kernel.
            assert self._depth is None, "depth supervision takes the separate kernels"
            self._render_deferred = False
            ra = _lib.RayTrainArgs()
            ra.density, ra.ebins, ra.sbins, ra.rgb = b["dens"][2].data_ptr(), b["eb"][2].data_ptr(), b["sb"][2].data_ptr(), b["rgb"].data_ptr()
            ra.bg, ra.target, ra.R, ra.S, ra.bg_mode = rng["bg"].data_ptr(), target.data_ptr(), R, S2, 0
            ra.go_scale, ra.dist_scale = 2.0 * co["rgb_loss"] / (3 * R), co["distortion_loss"] / R
            ra.weights, ra.rgb_out, ra.acc_out, ra.depth_median = b["w"][2].data_ptr(), b["rgb_out"].data_ptr(), b["acc"].data_ptr(), b["depth"].data_ptr()
            ra.sqerr_rays, ra.dist_rays, ra.g_rgb, ra.g_density = b["sqerr"].data_ptr(), b["dist_rays"].data_ptr(), b["grgb"].data_ptr(), b["gdens"][2].data_ptr()
            ra.g_weights, ra.nonfinite_flag = None, self._dyn["fields"].data_ptr()
            with self._span("ray_train_fwd_bwd"):
                _lib.check(self.lib.snerf_ray_train_fwd_bwd(C.byref(ra), self._st), "ray_train_fwd_bwd")
            if overlap and not sharded:
                proposal_chain()
        else:
            if overlap and not sharded:
                proposal_chain()
            # MSELoss (kplanes.py:418) folded into the render backward: g_rgb_out = 2 c / (3R) * (rgb_out - target); value lazily from sqerr
            _lib.check(self.lib.snerf_render_mse_bwd(self._p(b["w"][2]), self._p(b["rgb"]), self._p(rng["bg"]), 0, self._p(b["rgb_out"]), self._p(target),
                                                     2.0 * co["rgb_loss"] / (3 * R), R, S2, self._p(b["gw"][2]), self._p(b["grgb"]), self._p(b["sqerr"]),
                                                     self._st), "render_mse_bwd")
            _lib.check(self.lib.snerf_distortion(self._p(b["w"][2]), self._p(b["sb"][2]), R, S2, co["distortion_loss"] / R, self._p(b["dist_rays"]),
                                                 self._p(b["gw"][2]), 1, self._st), "distortion")
            self._depth_loss(2, with_grad=True)
            _lib.check(self.lib.snerf_weights_bwd(self._p(b["dens"][2]), self._p(b["eb"][2]), self._p(b["gw"][2]), R, S2, self._p(b["gdens"][2]), 0,
                                                  self._p(self._dyn["fields"]), self._st), "weights_bwd")
        if reg_done is not None:
            main.wait_event(reg_done)
        # the field chain stays on the caller's stream: every extra stream is one more HIP stream competing for the (four) hardware queues, and
        # two chains that land on one queue serialise (seen in the rocprofv3 timeline, profiles/r01_kernels.md)
        self._field_backward_chunk(0, R)
        if sharded:
            if not self._exchange_started:  # unsorted / deterministic scatter: every chunk is complete only now
                for k in range(len(self._exchange)):
                    self._start_field_grad_exchange(k)
            self._exchange_started = False
            if overlap:
                proposal_chain(after=main.record_event())
        if not overlap:
            self._proposal_backward(proposal_grads)
        # defer_prop_join (train_step, single GPU): the proposal chain may still be running when this returns -- nothing before the
        # proposal planes' own optimiser kernels needs its gradients, so its tail runs under the field planes' sweep instead of in
        # front of it; _join_prop() is the barrier
        self._prop_pending = None
        for st in joins:
            if defer_prop_join and not sharded and st is self._side.get("prop"):
                self._prop_pending = st
            else:
                main.wait_stream(st)
        # weight-gradient workspaces -> self.grads: the field's nets now, the proposal nets once their chain has been joined
        self._reduce_mlp_grads(("field.sigma", "field.color"))
        if self._prop_pending is None:
            self._reduce_mlp_grads(("prop0.mlp", "prop1.mlp"))

    def _join_prop(self):
        if self._prop_pending is not None:
            torch.cuda.current_stream().wait_stream(self._prop_pending)
            self._prop_pending = None
            self._reduce_mlp_grads(("prop0.mlp", "prop1.mlp"))

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        """Scaled loss terms of the last step, keys as KPlanesModel.get_loss_dict (kplanes.py:414-452).  Lazy: a few tiny
        reductions, only when asked for."""
        b, co, R = self.buf, self.cfg.loss_coefficients, self.R
        self._join_prop()
        if self._reg_work is not None:
            self._reg_work.wait()  # sharded optimiser: the field planes' regulariser values are summed across ranks asynchronously
        if self._field_adam_done is not None:
            torch.cuda.current_stream().wait_event(self._field_adam_done)  # the regulariser values come out of the (async) optimiser sweep
        d = dict(self.last)
        d["rgb_loss"] = b["sqerr"].sum() / (3 * R) * co["rgb_loss"]
        d["distortion_loss"] = b["dist_rays"].mean() * co["distortion_loss"]
        d["interlevel_loss"] = (b["inter_rays"][0].sum() + b["inter_rays"][1].sum()) / (R * self.S[2]) * co["interlevel_loss"]
        reg = b["reg"][:, :, :3].sum(1)  # [3 plane sets, 3 terms]
        d["space_tv_loss"], d["time_smoothness_loss"], d["sparse_transients_loss"] = (
            reg[0, 0] * co["space_tv_loss"], reg[0, 1] * co["time_smoothness_loss"], reg[0, 2] * co["sparse_transients_loss"])
        if self._depth is not None:
            d["depth_loss"] = sum(r.mean() for r in b["depth_rays"]) / 3 * co["depth_loss"]
        pr = reg[1] + reg[2]
        d["space_tv_proposal_loss"], d["time_smoothness_proposal_loss"], d["sparse_transients_proposal_loss"] = (
            pr[0] * co["space_tv_proposal_loss"], pr[1] * co["time_smoothness_proposal_loss"], pr[2] * co["sparse_transients_proposal_loss"])
        return d

    # ---- world > 1: reduce-scatter -> sharded Adam -> all-gather for the field planes (dist.py) ----
    def _sharded(self) -> bool:
        return self.world > 1 and self.shard_optimizer and self.fuse_reg_into_adam

    def _wait_params(self):
        """The current stream waits until the field planes of the last optimiser step are complete: the all-gather (sharded
        multi-GPU) or the sweep on the "adam" stream (async_field_adam).  No host block."""
        for ch in self._exchange:
            if ch["ag"] is not None:
                self._comm_wait(ch["ag"], "comm_wait.all_gather")
                ch["ag"] = None
                if self._delta_pending:
                    # bf16 parameter transport: what was gathered are the ranks' parameter UPDATES; every rank (the owner of a shard
                    # included) forms new = old + bf16(update), so the replicas stay bit-identical.  params / _params_alt were swapped since.
                    o = self._field_seg[0]
                    a, b = o + ch["lo"], o + ch["hi"]
                    torch.add(self._params_alt[a:b], ch["d16_full"], out=self.params[a:b])
        self._delta_pending = False
        ev = self._field_adam_done
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._field_adam_done = None

    def synchronize(self, check_overflow: bool = True):
        """Join every stream the trainer uses; call before reading parameters / Adam state from outside a train step."""
        self._join_prop()
        self._wait_params()
        torch.cuda.synchronize(self.dev)
        if check_overflow and getattr(self, "_fix_peak_host", None) is not None:
            self._ss.check_fix_overflow()

    @torch.no_grad()
    def restart(self, params: Optional[torch.Tensor] = None):
        """Back to optimiser step 0: Adam moments, gradients and the device-side step / skip counters cleared; `params` (a flat copy of
        self.params taken earlier, e.g. right after construction) restores the parameters too.  bench.py's trained-state leg starts from here."""
        self.synchronize(check_overflow=False)  # a new run starts here: an overflow recorded by the old one is cleared below, not raised
        if params is not None:
            self.params.copy_(params)
        for t in (self.exp_avg, self.exp_avg_sq, self.grads) + ((self.grads_fx,) if self.grads_fx is not None else ()):
            t.zero_()
        for d in self._dyn.values():
            d.zero_()
        self.step = self._dyn_step = 0
        self._steps_since_update = 0
        if getattr(self, "_fix_peak_host", None) is not None:  # a new run: the sticky overflow record of the old one is void
            self._ss.clear_fix_overflow()
            self._fix_peak_host.zero_()
        torch.cuda.synchronize(self.dev)

    def _start_field_grad_exchange(self, k: int):
        """Called on the stream that produced the gradient of exchange chunk k, right after its scatter: reduce-scatter(SUM) of the chunk
        into this rank's shard buffer, asynchronous on RCCL's stream."""
        from . import dist as sdist

        ch = self._exchange[k]
        o = self._field_seg[0]
        a, b = o + ch["lo"], o + ch["hi"]
        self._convert_fx(a, b)
        with self._span("reduce_scatter.field"):
            if self.grad_transport == "bf16":
                # opt-in: half the bytes on the links (each rank rounds its own gradient to bf16, the SUM is formed in bf16 by the collective);
                # NOT the reference's fp32 DDP all-reduce -- the optimiser then sees gradients with ~2^-9 relative rounding
                if ch["g16"] is None:
                    ch["g16"] = torch.empty(b - a, dtype=torch.bfloat16, device=self.dev)
                    ch["g16_shard"] = torch.empty(ch["shard"], dtype=torch.bfloat16, device=self.dev)
                ch["g16"].copy_(self.grads[a:b])
                ch["rs"] = sdist.reduce_scatter_sum(ch["g16_shard"], ch["g16"], self.pg, async_op=True)
            else:
                ch["rs"] = sdist.reduce_scatter_sum(ch["g_shard"], self.grads[a:b], self.pg, async_op=True)

    def _prepare_group(self, name: str, lr: float):
        """Once per step and parameter group, on the current stream, after the group's gradient producers and before its Adam kernels:
        skip decision + device-side step counter / bias corrections (ops.adam_prepare)."""
        if name in self._prepared:
            return
        if self._dyn_step != self.step:  # self.step was set from outside (checkpoint load, bench --start-step): Adam's counter follows
            for d in self._dyn.values():
                d[1] = self.step
            self._dyn_step = self.step
        if self.world > 1:  # DDP all-reduces the gradients, so a non-finite value on one rank is one on every rank: share the flag
            from . import dist as sdist

            with self._span("comm_wait.flags"):  # synchronous and tiny: its whole duration is link latency the chain waits for
                sdist.all_reduce_max_(self._dyn[name][:1], self.pg)
        ops.adam_prepare(self._dyn[name], lr, policy=self.cfg.nonfinite_policy)
        self._prepared.add(name)

    def _finish_step(self):
        self._prepared.clear()
        self.step += 1
        self._dyn_step = self.step
        self._poll_fix_overflow()

    def _poll_fix_overflow(self):
        """A fix list SMALLER than the worst case (cfg.fix_capacity set by the caller; the default cannot overflow) is watched from every optimiser
        step -- train_step's and a direct optimizer_step() caller's alike -- without ever blocking: every 8th step looks at the counter value copied
        into the pinned word 8 steps earlier and starts the next copy, so dropped gradient terms end the run within 16 steps (was: 128, and only
        through train_step).  synchronize() checks the device counter itself."""
        if self._fix_peak_host is None or self._ss.fix_capacity >= self._ss.N * self._ss.ps.out_dim or (self.step & 7) != 0:
            return
        self._ss.check_fix_overflow(int(self._fix_peak_host[0]))
        self._fix_peak_host.copy_(self._ss.fix_peak, non_blocking=True)

    def _convert_fx(self, lo: int = 0, hi: Optional[int] = None):
        """Deterministic mode: fixed-point cells [lo, hi) -> self.grads (added; the cells are cleared)."""
        if self.grads_fx is not None:
            hi = self.n_params if hi is None else hi
            ops.fx_to_float(self.grads_fx[lo:hi], self.grads[lo:hi], accumulate=True)

    def skipped_steps(self) -> Dict[str, int]:
        """Optimiser steps skipped so far per parameter group (non-finite gradients under nonfinite_policy = "skip_step") and
        gradient elements dropped by the Adam kernels.  Synchronises."""
        out = {}
        for name, d in self._dyn.items():
            h = d.cpu().tolist()
            out[name] = {"adam_steps": h[1], "skipped": h[2], "dropped_elements": h[3]}
        return out

    def _sharded_optimizer_step(self):
        """After backward() (all side streams joined): all-reduce of the small segments, Adam + regularisers on this rank's shard
        of the field planes, all-gather of the new planes (left in flight: forward() waits for it before the field gather)."""
        from . import dist as sdist

        cfg, co = self.cfg, self.cfg.loss_coefficients
        lr = cfg.lr * cosine_lr_factor(self.step, cfg.warm_up_end, cfg.max_steps, cfg.lr_alpha)
        gs = 1.0 / self.world
        off = {name: (o, n) for name, _, _, o, n in self.segments}
        o, n, npad = self._field_seg
        new = self._params_alt
        self._convert_fx(0, o)
        self._convert_fx(o + npad, None)
        self._prepare_group("fields", lr)
        self._prepare_group("proposal_networks", lr)
        dyn_f, dyn_p = self._dyn["fields"], self._dyn["proposal_networks"]
        # small segments: [prop0 planes | prop0 mlp | prop1 planes | prop1 mlp] before the field planes, [sigma | color] after them
        with self._span("allreduce_grads"):
            small = [self.grads[:o], self.grads[o + npad:]]
            self._ar_work = [sdist.all_reduce_sum_(t, self.pg, async_op=True) for t in small if t.numel()]
        if self._reg_work is not None:
            self._reg_work.wait()  # last step's regulariser-value reduction still reads buf["reg"]
        self.buf["reg"].zero_()
        coefs = tuple(co[k] for k in ("space_tv_loss", "time_smoothness_loss", "sparse_transients_loss"))
        n4 = _align4(n)
        for ch in self._exchange:  # in exchange order: the finest scale's shard is swept and gathered while the rest is still on the links
            lo = ch["lo"] + self.rank * ch["shard"]
            hi = min(lo + ch["shard"], n4)
            self._comm_wait(ch["rs"], "comm_wait.reduce_scatter")
            ch["rs"] = None
            self.grads[o + lo:o + lo + ch["shard"]].copy_(ch["g16_shard"] if self.grad_transport == "bf16" else ch["g_shard"])
            with self._span("adam_planes.field"):
                if hi > lo:
                    ops.adam_planes_step(self.field_planes, self.params[o:o + n], new[o:o + n], self.gviews["field.planes"], self.mviews["field.planes"],
                                         self.vviews["field.planes"], coefs, self.buf["reg"][0], self.step + 1, lr, eps=cfg.adam_eps, grad_scale=gs,
                                         zero_grad=False, shard_range=(lo, hi), dyn=dyn_f)
            if self.param_transport == "bf16":
                # opt-in: gather the shard's UPDATE in bf16 (half the bytes; 2^-9 relative rounding of the update, not of the parameter);
                # applied in _wait_params.  NOT the reference's semantics (replicas hold old + bf16(update) instead of the fp32 Adam result).
                if ch["d16_full"] is None:
                    ch["d16_full"] = torch.zeros(ch["hi"] - ch["lo"], dtype=torch.bfloat16, device=self.dev)
                    ch["d16_shard"] = torch.zeros(ch["shard"], dtype=torch.bfloat16, device=self.dev)
                ch["d16_shard"].zero_()
                if hi > lo:
                    ch["d16_shard"][:hi - lo].copy_(new[o + lo:o + hi] - self.params[o + lo:o + hi])
                with self._span("all_gather.field"):
                    ch["ag"] = sdist.all_gather_shards(ch["d16_full"], ch["d16_shard"], self.pg, async_op=True)
                self._delta_pending = True
            else:
                ch["p_shard"].copy_(new[o + lo:o + lo + ch["shard"]])
                with self._span("all_gather.field"):
                    ch["ag"] = sdist.all_gather_shards(new[o + ch["lo"]:o + ch["hi"]], ch["p_shard"], self.pg, async_op=True)
        for w in self._ar_work:
            self._comm_wait(w, "comm_wait.all_reduce")
        for i in range(2):
            name, ps = f"prop{i}.planes", self.prop_planes[i]
            with self._span(f"adam_planes.prop{i}"):
                ops.adam_planes_step(ps, self.views[name], new[off[name][0]:off[name][0] + off[name][1]], self.gviews[name], self.mviews[name],
                                     self.vviews[name], tuple(co[k] for k in ("space_tv_proposal_loss", "time_smoothness_proposal_loss",
                                                                              "sparse_transients_proposal_loss")),
                                     self.buf["reg"][1 + i], self.step + 1, lr, eps=cfg.adam_eps, grad_scale=gs, dyn=dyn_p)
        self._adam_mlps(new, off, lr, gs)
        self.grads[o:o + npad].zero_()  # the shard kernel leaves the gradient alone: clear the whole segment for the next step
        # regulariser VALUES of the field planes are per-shard partial sums: add them up across ranks (logging only)
        self._reg_work = sdist.all_reduce_sum_(self.buf["reg"][0], self.pg, async_op=True)
        self._params_alt = self.params
        self._repoint(new)
        self._finish_step()

    def _adam_mlps(self, new, off, lr, gs):
        with self._span("adam_step.mlps"):
            # MLP segments: prop0.mlp, prop1.mlp and the adjacent field.sigma + field.color
            o0, n0 = off["field.sigma"]
            o1, n1 = off["field.color"]
            dyn_f, dyn_p = self._dyn["fields"], self._dyn["proposal_networks"]
            for (o, n), dyn in ((off["prop0.mlp"], dyn_p), (off["prop1.mlp"], dyn_p), ((o0, o1 + n1 - o0), dyn_f)):
                n4 = (n + 3) // 4 * 4
                ops.adam_step(self.params[o:o + n4], self.grads[o:o + n4], self.exp_avg[o:o + n4], self.exp_avg_sq[o:o + n4], self.step + 1, lr,
                              eps=self.cfg.adam_eps, grad_scale=gs, zero_grad=True, p_out=new[o:o + n4], dyn=dyn)

    def _finest_offset(self) -> int:
        """First float of the finest scale's planes inside the field-plane segment (planes are laid out scale-major)."""
        return int(self._desc_field.off[len(self.cfg.multiscale_res) - 1][0])

    def _adam_field_range(self, lo: int, hi: Optional[int], side: bool, role: str = "adam", span: str = "adam_planes.field", after=None):
        """Fused Adam + regularisers over floats [lo, hi) of the field planes, old -> other half of the ping-pong pair.  side=True:
        on the side stream `role`, ordered after everything issued so far on the current stream (the scatter of those planes) and after
        the event `after` (pass B of those planes, when it ran on another stream)."""
        cfg, co = self.cfg, self.cfg.loss_coefficients
        lr = cfg.lr * cosine_lr_factor(self.step, cfg.warm_up_end, cfg.max_steps, cfg.lr_alpha)
        n4 = _align4(self.field_planes.numel)
        rng_ = (lo, n4 if hi is None else hi)
        o, n = next((o, n) for name, _, _, o, n in self.segments if name == "field.planes")
        args = (self.field_planes, self.params[o:o + n], self._params_alt[o:o + n], self.gviews["field.planes"], self.mviews["field.planes"],
                self.vviews["field.planes"], tuple(co[k] for k in ("space_tv_loss", "time_smoothness_loss", "sparse_transients_loss")),
                self.buf["reg"][0], self.step + 1, lr)
        self._prepare_group("fields", lr)
        kw = dict(eps=cfg.adam_eps, grad_scale=self._grad_scale, zero_grad=True, shard_range=rng_, dyn=self._dyn["fields"])
        if not side:
            with self._span(span):
                ops.adam_planes_step(*args, **kw)
            return
        cur = torch.cuda.current_stream()
        st = self._stream(role)
        st.wait_stream(cur)
        if after is not None:
            st.wait_event(after)
        with KPlanesTrainer._On(self, st), self._span(span):
            ops.adam_planes_step(*args, **kw)

    def allreduce_grads(self):
        """One all-reduce (SUM) over the flat gradient buffer; the mean (DDP semantics, base_pipeline.py:244-246) is folded
        into Adam's grad_scale.  RCCL when the group's backend is nccl (GPU), gloo in the CPU tests."""
        from . import dist as sdist

        if self.world > 1:
            self._join_prop()
            self._convert_fx()
        with self._span("allreduce_grads"):
            if self.cabi_comm is not None:  # the exchange behind the C ABI (snerf_allreduce_grads) instead of torch.distributed
                self._grad_scale = self.cabi_comm.all_reduce_sum_(self.grads)
            else:
                self._grad_scale = sdist.allreduce_flat_(self.grads, self.pg)

    def optimizer_step(self, fused_reg: bool = False):
        """Adam(lr*cosine, eps 1e-12) over the whole flat buffer + gradient clear (Optimizers.optimizer_step_all/scheduler_step_all)."""
        cfg, co = self.cfg, self.cfg.loss_coefficients
        lr = cfg.lr * cosine_lr_factor(self.step, cfg.warm_up_end, cfg.max_steps, cfg.lr_alpha)
        gs = self._grad_scale
        if self.grads_fx is not None:
            self._join_prop()
            self._convert_fx()
        if not fused_reg:
            self._join_prop()
            self._prepare_group("fields", lr)
            self._prepare_group("proposal_networks", lr)
            with self._span("adam_step"):  # one sweep per parameter group (= per torch optimiser of the reference)
                for lo, hi, g in ((0, self.n_proposal_params, "proposal_networks"), (self.n_proposal_params, self.n_params, "fields")):
                    ops.adam_step(self.params[lo:hi], self.grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.step + 1, lr, eps=cfg.adam_eps,
                                  grad_scale=gs, zero_grad=True, dyn=self._dyn[g])
            self._finish_step()
            return
        # regularisers fused into the sweep: plane sets go through snerf_adam_planes_step (values land in buf["reg"]), the MLP
        # segments through the plain kernel; everything writes the OTHER parameter buffer, which then becomes live
        new = self._params_alt
        off = {name: (o, n) for name, _, _, o, n in self.segments}
        sl = lambda t, name: t[off[name][0]:off[name][0] + off[name][1]]
        async_field = self.async_field_adam and self.overlap
        if not self._reg_zeroed:
            self.buf["reg"].zero_()
        # async: the big sweep goes to the "adam" stream and is NOT joined here -- the next step's pixel draw, ray generation and
        # proposal levels (which read only the small segments updated below) run under it; forward() joins before the field gather
        emu = self.cfg.emulate_transports if self.world == 1 else ""
        if emu in ("grad", "both"):
            gv = self.gviews["field.planes"]
            gv.copy_(gv.to(torch.bfloat16))
        if async_field and self._passb_fine_done is not None:
            # pipelined with pass B (cfg.pipeline_sweep): the planes of the scales scattered first on the "sort" stream (idle between this step's sort
            # and the next one's) as soon as their scatter is complete; the others' on the sweep's stream, behind the rest of pass B
            cut = self._finest_offset()
            ev, fine_first = self._passb_fine_done
            first, second = ((cut, None), (0, cut)) if fine_first else ((0, cut), (cut, None))
            names = ("adam_planes.field.fine", "adam_planes.field.coarse") if fine_first else ("adam_planes.field.coarse", "adam_planes.field.fine")
            self._adam_field_range(*first, side=True, role="sort", after=ev, span=names[0])
            fine_done = self._stream("sort").record_event()
            self._adam_field_range(*second, side=True, span=names[1])
            self._stream("adam").wait_event(fine_done)  # _field_adam_done (below) then stands for both
            self._passb_fine_done = None
            self.field_sweep_launches = 2
        else:
            self._passb_fine_done = None
            self._adam_field_range(0, None, side=async_field)
            self.field_sweep_launches = 1
        if emu in ("param", "both"):  # new = old + bf16(new - old) on the field planes, on the stream that ran the sweep
            o_, n_ = next((o, n) for name, _, _, o, n in self.segments if name == "field.planes")
            with KPlanesTrainer._On(self, self._stream("adam") if async_field else torch.cuda.current_stream()):
                old_, new_ = self.params[o_:o_ + n_], self._params_alt[o_:o_ + n_]
                new_.copy_(old_ + (new_ - old_).to(torch.bfloat16).float())
        if async_field:
            self._field_adam_done = self._stream("adam").record_event()
        self._reg_zeroed = False
        self._join_prop()  # the proposal gradients are needed from here on
        self._prepare_group("fields", lr)
        self._prepare_group("proposal_networks", lr)
        sets = [(f"prop{i}.planes", self.prop_planes[i], ("space_tv_proposal_loss", "time_smoothness_proposal_loss",
                                                          "sparse_transients_proposal_loss"), 1 + i) for i in range(2)]
        for name, ps, keys, row in sets:
            with self._span("adam_planes." + name.split(".")[0]):
                ops.adam_planes_step(ps, sl(self.params, name), sl(new, name), self.gviews[name], self.mviews[name], self.vviews[name],
                                     tuple(co[k] for k in keys), self.buf["reg"][row], self.step + 1, lr, eps=cfg.adam_eps, grad_scale=gs,
                                     dyn=self._dyn["proposal_networks"])
        self._adam_mlps(new, off, lr, gs)
        self._params_alt = self.params
        self._repoint(new)
        self._finish_step()

    def random_draws(self) -> Dict[str, torch.Tensor]:
        """The step's uniform draws (torch.rand on the current device stream), shaped as the reference's samplers draw them."""
        R, (S0, S1, S2) = self.R, self.S
        sj = self.cfg.use_single_jitter
        c0, c1, c2 = (1, 1, 1) if sj else (S0 + 1, S1 + 1, S2 + 1)
        flat = torch.rand(R * (c0 + c1 + c2 + 3), device=self.dev)
        o = 0
        out = []
        for c in (c0, c1, c2, 3):
            out.append(flat[o:o + R * c].view(R, c))
            o += R * c
        return {"t_rand": out[0], "u": [out[1], out[2]], "bg": out[3]}

    def train_step(self, rays: Dict[str, torch.Tensor], target: torch.Tensor, rng: Optional[Dict[str, torch.Tensor]] = None,
                   depth: Optional[torch.Tensor] = None):
        """One full training iteration (callbacks included): returns the rendered rgb [R,3] (a work buffer).  depth: termination depths [R]
        of the batch (batch["depth_image"]) when the dataset has depth maps."""
        cfg = self.cfg
        anneal = anneal_value(self.step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope)
        # the sampler's own step counter lags by one: it is set by the AFTER_TRAIN_ITERATION callback (kplanes.py:340-346)
        sstep = max(self.step - 1, 0)
        updated = self._steps_since_update > update_schedule(sstep, cfg.proposal_warmup, cfg.proposal_update_every) or sstep < 10
        rng = rng if rng is not None else self.random_draws()
        co = cfg.loss_coefficients
        defer = bool(cfg.fused_ray_loss and (depth is None or co.get("depth_loss", 0) <= 0))
        self._keep_pfeat = bool(updated)  # the proposal levels' features cross HBM only when their backward will read them
        try:
            out = self.forward(rays, rng, anneal, training=True, defer_render=defer)
        finally:
            self._keep_pfeat = True
        fuse = self.fuse_reg_into_adam
        self._in_train_step = True
        try:
            self.backward(target, rng, proposal_grads=updated, include_reg=not fuse,
                          defer_prop_join=fuse and self.world == 1 and self.defer_prop, depth=depth)
        finally:
            self._in_train_step = False
        if self._sharded():
            self._sharded_optimizer_step()
        else:
            self.allreduce_grads()
            self.optimizer_step(fused_reg=fuse)
        if updated:
            self._steps_since_update = 0
        self._steps_since_update += 1  # step_cb (ray_samplers.py:554-557)
        return out

    # ---- nerfstudio checkpoint files (trainer.py:331-380 of the reference; formats in soccernerfs_amd/checkpoint.py) ----
    def _named_module(self) -> torch.nn.Module:
        """The trainer's parameters under the reference model's module names (registration order = the reference's parameter order)."""
        nn = torch.nn
        root = nn.Module()
        aabb = lambda: nn.Parameter(torch.tensor(self.aabb, dtype=torch.float32), requires_grad=False)
        root.field = nn.Module()
        root.field.aabb = aabb()
        root.field.grids, root.field.sigma_net, root.field.color_net = self.field_planes, self.sigma_net, self.color_net
        root.proposal_networks = nn.ModuleList()
        for pp, pn in zip(self.prop_planes, self.prop_nets):
            m = nn.Module()
            m.aabb = aabb()
            m.grids, m.sigma_net = pp, pn
            root.proposal_networks.append(m)
        root.device_indicator_param = nn.Parameter(torch.empty(0))
        return root

    def _ck_names(self) -> Dict[str, str]:
        names = {"field.grids.planes": "field.planes", "field.sigma_net.params": "field.sigma", "field.color_net.params": "field.color"}
        for i in range(len(self.prop_planes)):
            names[f"proposal_networks.{i}.grids.planes"] = f"prop{i}.planes"
            names[f"proposal_networks.{i}.sigma_net.params"] = f"prop{i}.mlp"
        return names

    def _gather_moment_shards(self):
        """Sharded optimiser: every rank owns 1/world of each exchange chunk's Adam moments; make the full buffers whole on all ranks."""
        from . import dist as sdist

        off = self._field_seg[0]
        for buf in (self.exp_avg, self.exp_avg_sq):
            for ch in self._exchange:
                seg = buf[off + ch["lo"]:off + ch["hi"]]
                sdist.all_gather_shards(seg, seg[self.rank * ch["shard"]:(self.rank + 1) * ch["shard"]].clone(), self.pg)

    def save_checkpoint(self, checkpoint_dir: str, save_only_latest_checkpoint: bool = True) -> Optional[str]:
        """Writes `step-%09d.ckpt` (rank 0 only; collective when the optimiser is sharded).  The saved step is the index of the last
        completed iteration, as the reference saves it; load_checkpoint resumes at step + 1."""
        from . import checkpoint as CK

        self.synchronize()
        if self._sharded():
            self._gather_moment_shards()
        if self.rank != 0:
            return None
        moments = {ck: (self.mviews[seg], self.vviews[seg], self.step) for ck, seg in self._ck_names().items()} if self.step > 0 else {}
        root = self._named_module()
        hyper = {"lr": self.cfg.lr, "betas": (0.9, 0.999), "eps": self.cfg.adam_eps, "weight_decay": 0, "amsgrad": False}
        opt = CK.export_optimizer_states(root, moments, {"fields": hyper, "proposal_networks": hyper})
        return CK.save_checkpoint(checkpoint_dir, max(self.step - 1, 0), root, opt, save_only_latest_checkpoint)

    def load_checkpoint(self, load_dir: str, load_step: Optional[int] = None) -> int:
        """Parameters (+ Adam moments when the file holds them) from a nerfstudio checkpoint; returns the step training resumes at."""
        from . import checkpoint as CK

        self.synchronize()
        root = self._named_module()
        start, moments = CK.load_checkpoint(load_dir, root, load_step)
        names = self._ck_names()
        for ck, (m, v, _) in moments.items():
            self.mviews[names[ck]].copy_(m.reshape(-1).to(self.dev))
            self.vviews[names[ck]].copy_(v.reshape(-1).to(self.dev))
        self._params_alt.copy_(self.params)
        self.step = start if moments else 0  # Adam bias correction counts optimiser updates; without moments the optimiser restarts
        self._steps_since_update = 0
        torch.cuda.synchronize(self.dev)
        return start

    # ---- reference-layout import (parity tests / checkpoint import) ----
    @torch.no_grad()
    def load_oracle_params(self, P: Dict):
        self.field_planes.load_reference([[t.to(self.dev) for t in sc] for sc in P["field_grids"]])
        self.sigma_net.load_linear_weights([w.to(self.dev) for w in P["field_sigma"]])
        self.color_net.load_linear_weights([w.to(self.dev) for w in P["field_color"]])
        for i in range(len(self.prop_planes)):
            self.prop_planes[i].load_reference([[t.to(self.dev) for t in P["prop_grids"][i]]])
            self.prop_nets[i].load_linear_weights([w.to(self.dev) for w in P["prop_sigma"][i]])
