"""Fused training step of the reference's NeRFPlayer-nerfacto model (config 4) on libsnerf: one flat parameter / gradient / Adam
buffer, preallocated per-sample work buffers, no autograd graph and no host synchronisation inside a step (the only tensors created
per step are three per-RAY temporaries -- SH of the directions, the gathered appearance rows and their summed gradient -- from
torch's caching allocator).

What the reference does per step: Trainer.train_iteration -> NerfplayerNerfactoModel.get_outputs / get_metrics_dict / get_loss_dict
(NS/models/nerfplayer_nerfacto.py:206-318) -> autograd backward -> 2x Adam (NS/configs/method_configs.py:648-657: lr 1e-2,
eps 1e-12, cosine schedule with 512 warm-up steps) -> callbacks (proposal-weight annealing, proposal update schedule:
NS/models/nerfacto.py:235-264).  Same mathematics here, in this order:

  collider (AABB) -> piecewise sampler (single jitter) -> [temporal hash grid -> 10->16->1 MLP -> trunc_exp -> weights -> PDF] x2
  -> main grid (16 levels) -> 32->64->16 MLP (density = trunc_exp(col 0)) -> [SH4(dir) | 15 geo features | appearance(cam)] ->
  63->64->64->3 sigmoid MLP -> weights -> rgb / accumulation / expected depth -> MSE + interlevel + 1e-3 distortion + temporal TV x3
  -> gradients of all of it -> one Adam sweep (gradient cleared in the sweep) with the TV gradient added just before it.

The nerfstudio-shaped autograd model (nerfplayer_nerfacto.py, pinned against the reference by golden G12) is the checker of this file:
tests/test_gpu_nerfplayer_trainer.py compares losses and every gradient tensor on identical draws."""
import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, ops
from .streams import side_stream
from .nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
from .tcnn_compat import Network
from .temporal_grid import TemporalGridEncoder
from .trainer import anneal_value, cosine_lr_factor


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class NerfplayerTrainer:
    def __init__(self, cfg: NerfplayerNerfactoModelConfig, num_rays: int, num_images: int, aabb_scale: float = 1.0, device="cuda:0",
                 lr: float = 1e-2, adam_eps: float = 1e-12, warm_up_end: int = 512, max_steps: int = 30000, seed: int = 0, deterministic: bool = False,
                 async_field_sweep: bool = False, mlp_operands: str = "fp32", tiled_field_backward: bool = False, tiled_first_level: int = 0):
        """mlp_operands: "fp32" (exact: the parity path, what G12 pins) or "bf16" (bf16 MFMA operands, fp32 accumulation, for every net whose shape the
        16-bit fused kernels are built for -- the decode net and the colour head of the preset; the reference runs all of them in tcnn's fp16).
        async_field_sweep (round 5): inside train_step the optimiser sweep of the FIELD's table (most of the parameters) is launched on a side stream
        as soon as that table's gradient is complete (right behind the field's temporal-grid backward) and is joined only in front of the NEXT forward's
        field level: it runs beside the proposal networks' backward (atomic-bound) and the next step's ray generation and proposal levels, which read
        the two small tables only.  Same arithmetic, same bits in deterministic mode.  Readers of the field table outside forward() call wait_params() /
        synchronize() first; off by default for that reason (bench.py's config-4 leg and tools/bench_nerfplayer.py turn it on).
        tiled_field_backward (round 6): inside train_step the FIELD table's gradient scatter and its Adam sweep are one owner-computes pass
        (temporal_grid.TiledTableBackward, csrc/tgrid_tiles.hip): the batch's (sample, level, corner) touches are binned by tile of 256 table rows, one
        workgroup per tile sums its rows in LDS and steps them with Adam straight from there -- no float atomics (the run-length kernel ran AT the chip's
        atomic rate: 1.26 ms per step on camera rays) and no dense gradient for this table (24 instead of 32 B per parameter in the sweep).  Same
        mathematics; float sums in another association order.  The fused pass goes where the asynchronous sweep goes (side stream with
        async_field_sweep, else the caller's stream).  backward() called outside train_step, and deterministic mode, keep the atomic scatter into
        self.grads.  tiled_first_level: levels below it stay with the atomic kernel (0: every level tiled).
        deterministic: every gradient scatter (temporal-grid tables, MLP weight gradients, appearance embedding) accumulates 2^50-scaled 64-bit
        integers instead of float atomics (csrc/common.hpp: integer addition is associative), converted once per step: two runs from the same seed give
        the same bits.  Costs 8 B per parameter and 64-bit atomics; off by default."""
        if not cfg.disable_scene_contraction or cfg.use_same_proposal_network or cfg.num_proposal_iterations != 2:
            raise NotImplementedError("NerfplayerTrainer covers the nerfplayer-nerfacto preset (AABB collider, two proposal networks)")
        self.cfg, self.R, self.dev = cfg, num_rays, torch.device(device)
        self.lr, self.adam_eps, self.warm_up_end, self.max_steps = lr, adam_eps, warm_up_end, max_steps
        a = aabb_scale
        self.aabb = [[-a, -a, -a], [a, a, a]]
        torch.manual_seed(seed)
        if mlp_operands not in ("fp32", "bf16"):
            raise ValueError(f"mlp_operands must be 'fp32' or 'bf16', got {mlp_operands!r}")
        self.mlp_operands = mlp_operands

        def mlp(din, dout, h, nh, act):
            d = _lib.MlpDesc()
            d.d_in, d.d_out, d.hidden, d.n_hidden, d.hidden_act, d.out_act, d.operands = din, dout, h, nh, 1, int(act == "Sigmoid"), int(mlp_operands == "bf16")
            op = mlp_operands if (d.operands and _lib.lib().snerf_mlp_supported(C.byref(d))) else "fp32"
            return Network(din, dout, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": act, "n_neurons": h, "n_hidden_layers": nh}, operands=op)

        # ---- modules exactly as the fields build them (nerfplayer_nerfacto_field.py:83-104, 238-311) ----
        self.prop_enc: List[TemporalGridEncoder] = []
        self.prop_mlp: List[Network] = []
        for args in cfg.proposal_net_args_list[:2]:
            L, H = args.get("num_levels", 8), args.get("hidden_dim", 64)
            growth = float(np.exp((np.log(args.get("max_res", 1024)) - np.log(16)) / (L - 1)))
            self.prop_enc.append(TemporalGridEncoder(input_dim=3, temporal_dim=args.get("temporal_dim", 64), num_levels=L, level_dim=2,
                                                     per_level_scale=growth, base_resolution=16, log2_hashmap_size=args.get("log2_hashmap_size", 18)))
            self.prop_mlp.append(mlp(2 * L, 1, H, 1, "None"))
        self.enc = TemporalGridEncoder(input_dim=3, temporal_dim=cfg.temporal_dim, num_levels=cfg.num_levels, level_dim=cfg.features_per_level,
                                       log2_hashmap_size=cfg.log2_hashmap_size, desired_resolution=1024 * 2.0 * a)
        self.decode = mlp(cfg.num_levels * cfg.features_per_level, 16, 64, 1, "None")
        self.head = mlp(16 + 15 + 32, 3, 64, 2, "Sigmoid")
        self.appearance = torch.nn.Embedding(num_images, 32)
        # ---- one flat buffer ----
        self.segments = []  # (name, tensor-owner, attr, offset, numel)
        off = 0
        for i in range(2):
            for name, mod, attr in ((f"prop{i}.table", self.prop_enc[i], "embeddings"), (f"prop{i}.mlp", self.prop_mlp[i], "params")):
                n = getattr(mod, attr).numel()
                self.segments.append((name, mod, attr, off, n))
                off += _align4(n)
        for name, mod, attr in (("field.table", self.enc, "embeddings"), ("field.decode", self.decode, "params"), ("field.head", self.head, "params"),
                                ("field.appearance", self.appearance, "weight")):
            n = getattr(mod, attr).numel()
            self.segments.append((name, mod, attr, off, n))
            off += _align4(n)
        self.n_params = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.grads_fx = torch.zeros(off, dtype=torch.int64, device=self.dev) if deterministic else None
        self.views, self.gviews, self.fxviews = {}, {}, {}
        for name, mod, attr, o, n in self.segments:
            p = getattr(mod, attr)
            self.params[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)  # the module's parameter aliases its segment
            self.views[name] = p.data
            self.gviews[name] = self.grads[o:o + n].view(p.shape)
            if deterministic:
                self.fxviews[name] = self.grads_fx[o:o + n].view(p.shape)
        # ---- work buffers ----
        R = num_rays
        S0, S1 = cfg.num_proposal_samples_per_ray
        S2 = cfg.num_nerf_samples_per_ray
        self.S = (S0, S1, S2)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
        self.buf = {
            "sb": [f(R, s + 1) for s in self.S], "eb": [f(R, s + 1) for s in self.S],
            "dens": [f(R, s) for s in self.S], "w": [f(R, s) for s in self.S], "gw": [f(R, s) for s in self.S], "gdens": [f(R, s) for s in self.S],
            "pfeat": [f(R * S0, self.prop_enc[0].output_dim), f(R * S1, self.prop_enc[1].output_dim)],
            "pout": [f(R * S0, 1), f(R * S1, 1)],
            "gpfeat": [f(R * S0, self.prop_enc[0].output_dim), f(R * S1, self.prop_enc[1].output_dim)],
            "feat": f(R * S2, self.enc.output_dim), "gfeat": f(R * S2, self.enc.output_dim),
            "h": f(R * S2, 16), "gh": z(R * S2, 16),
            "hx": z(R * S2, 64), "ghx": f(R * S2, 64),           # head input [SH 16 | geo 15 | appearance 32 | pad], and its gradient
            "rgb": f(R * S2, 3), "grgb": f(R * S2, 3),
            "rgb_out": f(R, 3), "acc": f(R), "depth": f(R), "sqerr": z(R), "dist_rays": f(R), "inter_rays": [f(R), f(R)],
            "tv": z(3, 64, 16),
        }
        self._srow = [z(e.embeddings.shape[0]) for e in (self.enc, self.prop_enc[0], self.prop_enc[1])]  # per-row TV step of each table
        self.lib = _lib.lib()
        self.step = 0
        self._steps_since_update = 0
        self._timing, self._timing_all = None, False
        self.tv_rows: Optional[List[int]] = None  # parity hook: fixed table rows [field, prop0, prop1] instead of the random draw
        self._tv_cols = [(0, 1)] * 3
        self.async_field_sweep = bool(async_field_sweep)
        self.tiled_field_backward = bool(tiled_field_backward) and not deterministic
        self._tiled, self._bin_done, self.early_bin = None, None, True
        if self.tiled_field_backward:
            from .temporal_grid import TiledTableBackward

            self._tiled = TiledTableBackward(self.enc, R * S2, first_tiled_level=tiled_first_level)
        self._side, self._field_sweep_done, self._field_swept, self._in_train_step, self._tv0_done = None, None, False, False, None
        self._tv12_done = None
        self.fused_ray_loss, self._ray_deferred = True, False  # train_step: the nerf level's per-ray work as one launch (snerf_ray_train_fwd_bwd)

    # ---- helpers ----
    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    # ---- HIP events around kernel groups (bench.py's config-4 leg; same contract as KPlanesTrainer.enable_kernel_timing) ----
    def enable_kernel_timing(self, names=None):
        self._timing = {} if names is None else {n: [] for n in names}
        self._timing_all = names is None

    def disable_kernel_timing(self):
        self._timing = None

    def kernel_times_ms(self):
        """name -> (mean milliseconds per launch, launches).  Synchronises."""
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in evs) / len(evs), len(evs)) for k, evs in (self._timing or {}).items() if evs}

    def _span(self, name):
        from .trainer import KPlanesTrainer

        return KPlanesTrainer._Span(self, name)

    def wait_params(self):
        """The current stream waits for the field table's asynchronous sweep (no host block)."""
        # the event is KEPT until the next sweep replaces it: waiting for a completed event is free, and a later reader on ANOTHER stream (a checkpoint
        # save, a side-stream evaluation) that calls wait_params() is then ordered behind the sweep too
        if self._field_sweep_done is not None:
            torch.cuda.current_stream().wait_event(self._field_sweep_done)

    def synchronize(self):
        self.wait_params()
        torch.cuda.synchronize()

    def _tv_sign(self, k: int):
        """Value + per-row signed step of table k's temporal TV (k: 0 field, 1 / 2 proposal tables); the row draw is the reference's randint."""
        cfg, enc = self.cfg, (self.enc, self.prop_enc[0], self.prop_enc[1])[k]
        row = self.tv_rows[k] if self.tv_rows is not None else int(torch.randint(0, len(enc._index_list_host), [1]).item())
        ca, cb = enc._index_list_host[row]
        self._tv_cols[k] = (ca, cb)
        rows_, gc = enc.embeddings.shape
        _lib.check(self.lib.snerf_tgrid_tv_sign(self._p(enc.embeddings), C.c_int64(rows_), gc, ca, cb, float(cfg.temporal_tv_weight),
                                                self._p(self.buf["tv"][k]), 64, self._p(self._srow[k]), self._st), "tv_sign")

    def _sweep_table(self, name: str, lr: float, st):
        """Adam over one table on the current stream (st = its handle), with the temporal-TV gradient of its two columns added on the fly."""
        o, n = next((o, n) for nm, _, _, o, n in self.segments if nm == name)
        sl = slice(o, o + n)
        if self.cfg.temporal_tv_weight <= 0:
            ops.adam_step(self.params[sl], self.grads[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.step + 1, lr, eps=self.adam_eps, zero_grad=True)
            return
        kk = {"field.table": 0, "prop0.table": 1, "prop1.table": 2}[name]
        enc = self.enc if kk == 0 else self.prop_enc[kk - 1]
        ca, cb = self._tv_cols[kk]
        rows_, gc = enc.embeddings.shape
        with self._span("adam_tv." + name):
            _lib.check(self.lib.snerf_adam_step_tv(self._p(self.params[sl]), self._p(self.grads[sl]), self._p(self.exp_avg[sl]), self._p(self.exp_avg_sq[sl]),
                                                   C.c_int64(rows_), gc, ca, cb, self._p(self._srow[kk]), lr, 0.9, 0.999, self.adam_eps, self.step + 1, 1.0, 1,
                                                   None, st), "adam_step_tv")

    def _field_tv_early(self):
        """async_field_sweep: the field table's TV pass reads parameters only, so it goes to the side stream at the START of the backward instead of between the
        table's gradient scatter and its sweep (~0.1 ms off the chain scatter -> sweep -> next forward)."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = side_stream(self.dev, "adam")  # the process-wide sweep stream (streams.py)
        self._side.wait_stream(main)  # b["tv"] zeroed; the forward's reads of everything else are irrelevant to it
        keep = self._st
        with torch.cuda.stream(self._side):
            self._st = C.c_void_p(self._side.cuda_stream)
            try:
                self._tv_sign(0)
            finally:
                self._st = keep
            self._tv0_done = self._side.record_event()
        # (r06) the two proposal tables' TV passes read parameters only as well: on the "sort" stream now, beside the field's backward, instead of behind the
        # proposal backward on the caller's stream -- where they shared the memory system with the tile pass / sweep of the main table (0.5 ms of column reads
        # over the tables inside the window of the step's critical kernel).  Same order of the reference's row draws: field, proposal 0, proposal 1.
        sb = side_stream(self.dev, "sort")
        sb.wait_stream(main)
        with torch.cuda.stream(sb):
            self._st = C.c_void_p(sb.cuda_stream)
            try:
                self._tv_sign(1)
                self._tv_sign(2)
            finally:
                self._st = keep
            self._tv12_done = sb.record_event()

    def _field_table_sweep_async(self):
        """async_field_sweep: the field table's Adam sweep on the side stream, behind everything the caller's stream holds (the table's gradient scatter)
        and behind its TV pass (same stream)."""
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)  # (the table's TV pass has been on this stream since the start of the backward: _field_tv_early)
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
        with torch.cuda.stream(self._side):
            if self.grads_fx is not None:
                gv = self.gviews["field.table"]
                ops.fx_to_float(self._fx_of(gv), gv.view(-1), accumulate=True)
            self._sweep_table("field.table", lr, C.c_void_p(self._side.cuda_stream))
            self._field_sweep_done = self._side.record_event()
        self._field_swept = True

    def _field_table_fused_adam(self, on_side: bool):
        """tiled_field_backward: the field table's scatter + temporal TV + Adam as one owner-computes pass over the tiles filed by `bin`; on the side stream
        (async_field_sweep) it reads only the tiler's own buffers and gfeat, which the next step rewrites after wait_params() at the earliest."""
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
        o, n = next((o, n) for nm, _, _, o, n in self.segments if nm == "field.table")
        sl = slice(o, o + n)
        tv = self._tv_cols[0] if self.cfg.temporal_tv_weight > 0 else None
        gt = self.gviews["field.table"] if self._tiled.plan.first_tiled_level > 0 else None

        def run(st):
            with self._span("tgrid_tiles_adam.field"):
                self._tiled.scatter_adam(self.buf["gfeat"], gt, self.params[sl], self.exp_avg[sl], self.exp_avg_sq[sl], lr, self.step + 1, self.adam_eps, tv_cols=tv,
                                         srow=self._srow[0] if tv is not None else None, stream=st)

        if on_side:
            if self._side is None:
                self._side = side_stream(self.dev, "adam")
            self._side.wait_stream(torch.cuda.current_stream())  # (the table's TV pass has been on this stream since the start of the backward)
            with torch.cuda.stream(self._side):
                run(C.c_void_p(self._side.cuda_stream))
                self._field_sweep_done = self._side.record_event()
        else:
            run(self._st)
        self._field_swept = True

    def _tgrid_fwd(self, enc, table, co, times, S, N, out):
      with self._span("tgrid_fwd.field" if enc is self.enc else "tgrid_fwd.prop"):
        _lib.check(self.lib.snerf_tgrid_encode_fwd(C.byref(enc.desc), self._p(table), C.byref(co), None, self._p(times), S, C.c_int64(N), self._p(out),
                                                   self._st), "tgrid_fwd")

    def _fx_of(self, gview: torch.Tensor) -> torch.Tensor:
        """The fixed-point cells behind a view of self.grads (deterministic mode)."""
        o = gview.storage_offset() - self.grads.storage_offset()
        return self.grads_fx[o:o + gview.numel()]

    def _tgrid_bwd(self, enc, co, times, S, N, gout, gtable):
      with self._span("tgrid_bwd.field" if enc is self.enc else "tgrid_bwd.prop"):
        if self.grads_fx is not None:
            _lib.check(self.lib.snerf_tgrid_encode_bwd_fx(C.byref(enc.desc), C.byref(co), None, self._p(times), S, C.c_int64(N), self._p(gout),
                                                          self._p(self._fx_of(gtable)), self._st), "tgrid_bwd_fx")
            return
        _lib.check(self.lib.snerf_tgrid_encode_bwd(C.byref(enc.desc), C.byref(co), None, self._p(times), S, C.c_int64(N), self._p(gout), self._p(gtable),
                                                   self._st), "tgrid_bwd")

    def _mlp_fwd(self, net, X, ldx, N, Y, ldy, aux_col=-1, aux=None):
        _lib.check(self.lib.snerf_mlp_fwd(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(Y), ldy, aux_col,
                                          self._p(aux) if aux is not None else None, self._st), "mlp_fwd")

    def _mlp_bwd(self, net, gW, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx):
        if self.grads_fx is not None:
            _lib.check(self.lib.snerf_mlp_bwd_fx(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(gY) if gY is not None else None,
                                                 ldgy, aux_col, self._p(gaux) if gaux is not None else None, self._p(gX) if gX is not None else None, ldgx,
                                                 self._p(self._fx_of(gW)), self._st), "mlp_bwd_fx")
            return
        _lib.check(self.lib.snerf_mlp_bwd(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(gY) if gY is not None else None,
                                          ldgy, aux_col, self._p(gaux) if gaux is not None else None, self._p(gX) if gX is not None else None, ldgx,
                                          self._p(gW), self._st), "mlp_bwd")

    def _resample(self, lvl, rand, anneal):
        b, a = self.buf, _lib.ResampleArgs()
        a.density, a.ebins_prev, a.weights_out = b["dens"][lvl].data_ptr(), b["eb"][lvl].data_ptr(), b["w"][lvl].data_ptr()
        a.sbins_prev, a.nears, a.fars = b["sb"][lvl].data_ptr(), self.rays["nears"].data_ptr(), self.rays["fars"].data_ptr()
        if rand is None:
            a.u_mode = 2
        else:
            a.u_mode, a.u_or_rand, a.rand_cols = 1, rand.data_ptr(), rand.shape[-1]
        a.sbins_out, a.ebins_out = b["sb"][lvl + 1].data_ptr(), b["eb"][lvl + 1].data_ptr()
        a.R, a.S_prev, a.S, a.kind = self.R, self.S[lvl], self.S[lvl + 1], 1  # spacing = UniformLinDispPiecewise (ray_samplers.py:242-243)
        a.anneal, a.histogram_padding, a.eps = anneal, 0.01, 1e-5
        _lib.check(self.lib.snerf_pdf_resample(C.byref(a), self._st), "pdf_resample")

    # ---- forward ----
    def forward(self, rays: Dict[str, torch.Tensor], cams: Optional[torch.Tensor], rng: Dict[str, torch.Tensor], anneal: float, training: bool = True):
        """rays: origins [R,3], directions [R,3] (unit), times [R,1]; cams int64 [R]; rng: t_rand [R,1], u [2 x [R,1]], bg [R,3].
        training=False (exactly R rays): no jitter, deterministic PDF samples, average / zero appearance embedding
        (nerfplayer_nerfacto_field.py:362-372), eval-mode compositing; rng needs only bg (the "random" background is drawn in eval too,
        renderers.py:102-104)."""
        cfg, b, R = self.cfg, self.buf, self.R
        self._st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        o, d, t = rays["origins"], rays["directions"], rays["times"].reshape(-1)
        rays = dict(rays)
        rays["nears"], rays["fars"] = ops.aabb_collide(o, d, self.aabb, 0.0, training)  # AABBBoxCollider(scene_box): near_plane 0
        if cams is not None and (cams.dtype != torch.int64 or not cams.is_contiguous()):
            cams = cams.to(torch.int64).contiguous()
        self.rays, self.cams = rays, cams
        t_rand = rng["t_rand"] if training else None
        _lib.check(self.lib.snerf_spaced_bins(self._p(rays["nears"]), self._p(rays["fars"]), self._p(t_rand) if t_rand is not None else None,
                                              t_rand.shape[-1] if t_rand is not None else 0, R,
                                              self.S[0], 1, self._p(b["sb"][0]), self._p(b["eb"][0]), self._st), "spaced_bins")
        self._coords = []
        for lvl in range(3):
            co = ops.coords_from_rays(o, d, t, b["eb"][lvl], self.aabb, False)
            self._coords.append(co)
            S, N = self.S[lvl], R * self.S[lvl]
            if lvl < 2:
                enc, net = self.prop_enc[lvl], self.prop_mlp[lvl]
                self._tgrid_fwd(enc, enc.embeddings, co, t, S, N, b["pfeat"][lvl])
                self._mlp_fwd(net, b["pfeat"][lvl], enc.output_dim, N, b["pout"][lvl], 1, 0, b["dens"][lvl])
                self._resample(lvl, rng["u"][lvl] if training else None, anneal)
            else:
                self.wait_params()  # the field table's sweep of the last step (async_field_sweep) must be complete before the table is read
                if self._tiled is not None and self._in_train_step and training and self.early_bin:
                    # the binning pass of the tiled backward needs the sample positions only: on a side stream NOW, beside the field forward and the MLP
                    # backward, instead of between the decode net's backward and the tile pass on the critical chain (~0.19 ms).  Behind wait_params(): the
                    # previous step's tile pass has read the tiler's buffers by then.
                    main = torch.cuda.current_stream()
                    sb = side_stream(self.dev, "sort")
                    sb.wait_stream(main)
                    with torch.cuda.stream(sb), self._span("tgrid_bin.field"):
                        self._tiled.bin(co, t, S, None, C.c_void_p(sb.cuda_stream))
                        self._bin_done = sb.record_event()
                self._tgrid_fwd(self.enc, self.enc.embeddings, co, t, S, N, b["feat"])
                self._mlp_fwd(self.decode, b["feat"], self.enc.output_dim, N, b["h"], 16, 0, b["dens"][2])
                # colour head input [SH 16 | geo 15 | appearance 32 | 0] in one launch (csrc/nerfplayer.hip; ~35 ATen kernels before round 5)
                if training:
                    app, cm = self.appearance.weight, cams
                elif cfg.use_average_appearance_embedding:
                    app, cm = self.appearance.weight.mean(0, keepdim=True).contiguous(), None
                else:
                    app, cm = None, None
                dd = d if d.is_contiguous() else d.contiguous()
                _lib.check(self.lib.snerf_nerfacto_head_input_fwd(self._p(dd), self._p(b["h"]), self._p(app) if app is not None else None,
                                                                  self._p(cm) if cm is not None else None, S, R, self._p(b["hx"]), self._st), "head_input_fwd")
                self._mlp_fwd(self.head, b["hx"], 64, N, b["rgb"], 3)
                self._ray_deferred = bool(training and self._in_train_step and self.fused_ray_loss)
                if not self._ray_deferred:
                    _lib.check(self.lib.snerf_weights_fwd(self._p(b["dens"][2]), self._p(b["eb"][2]), R, S, self._p(b["w"][2]), self._st), "weights_fwd")
        if self._ray_deferred:
            # train_step: weights, compositing, MSE backward, distortion and the weights' backward are ONE launch at the start of backward()
            # (snerf_ray_train_fwd_bwd, bit-identical to the five kernels); the expected depth (an output no loss of this model reads) is not rendered
            return b["rgb_out"]
        a = _lib.RenderArgs()
        a.weights, a.rgb, a.ebins = b["w"][2].data_ptr(), b["rgb"].data_ptr(), b["eb"][2].data_ptr()
        a.bg_mode, a.bg = 0, rng["bg"].data_ptr()
        a.R, a.S, a.training = R, self.S[2], int(training)
        a.rgb_out, a.acc_out, a.depth_expected = b["rgb_out"].data_ptr(), b["acc"].data_ptr(), b["depth"].data_ptr()
        _lib.check(self.lib.snerf_render_fwd(C.byref(a), self._st), "render_fwd")
        return b["rgb_out"]

    # ---- backward ----
    def backward(self, target: torch.Tensor, rng: Dict[str, torch.Tensor], proposal_grads: bool):
        cfg, b, R = self.cfg, self.buf, self.R
        S2, N2 = self.S[2], R * self.S[2]
        t = self.rays["times"].reshape(-1)
        target = target if target.is_contiguous() else target.contiguous()
        # a backward that raised after its asynchronous sweep was issued never reached optimizer_step(): start from clean flags, or this step's TV pass and
        # sweep of the field table would be skipped
        self._field_swept, self._tv0_done = False, None
        early = bool(self.async_field_sweep and self._in_train_step)
        if cfg.temporal_tv_weight > 0:
            b["tv"].zero_()
            if early:
                self._field_tv_early()
        if self._ray_deferred:
            ra = _lib.RayTrainArgs()
            ra.density, ra.ebins, ra.sbins, ra.rgb = b["dens"][2].data_ptr(), b["eb"][2].data_ptr(), b["sb"][2].data_ptr(), b["rgb"].data_ptr()
            ra.bg, ra.target, ra.R, ra.S, ra.bg_mode = rng["bg"].data_ptr(), target.data_ptr(), R, S2, 0
            ra.go_scale, ra.dist_scale = 2.0 / (3 * R), cfg.distortion_loss_mult / R
            ra.weights, ra.rgb_out, ra.acc_out, ra.depth_median = b["w"][2].data_ptr(), b["rgb_out"].data_ptr(), b["acc"].data_ptr(), None
            ra.sqerr_rays, ra.dist_rays, ra.g_rgb, ra.g_density = b["sqerr"].data_ptr(), b["dist_rays"].data_ptr(), b["grgb"].data_ptr(), b["gdens"][2].data_ptr()
            ra.g_weights, ra.nonfinite_flag = None, None
            _lib.check(self.lib.snerf_ray_train_fwd_bwd(C.byref(ra), self._st), "ray_train_fwd_bwd")
            self._ray_deferred = False
        else:
            _lib.check(self.lib.snerf_render_mse_bwd(self._p(b["w"][2]), self._p(b["rgb"]), self._p(rng["bg"]), 0, self._p(b["rgb_out"]), self._p(target),
                                                     2.0 / (3 * R), R, S2, self._p(b["gw"][2]), self._p(b["grgb"]), self._p(b["sqerr"]), self._st), "render_mse_bwd")
            _lib.check(self.lib.snerf_distortion(self._p(b["w"][2]), self._p(b["sb"][2]), R, S2, cfg.distortion_loss_mult / R, self._p(b["dist_rays"]),
                                                 self._p(b["gw"][2]), 1, self._st), "distortion")
            _lib.check(self.lib.snerf_weights_bwd(self._p(b["dens"][2]), self._p(b["eb"][2]), self._p(b["gw"][2]), R, S2, self._p(b["gdens"][2]), 0, None,
                                                  self._st), "weights_bwd")
        # colour head: gX = [dSH (unused) | d geo | d appearance | pad]
        self._mlp_bwd(self.head, self.gviews["field.head"], b["hx"], 64, N2, b["grgb"], 3, -1, None, b["ghx"], 64)
        # gh[:, 1:16] = ghx[:, 16:31] (column 0, the density, enters through gaux below); appearance gradient = per-ray sums of ghx[:, 31:63] added to the
        # cameras' rows -- fixed-point cells in deterministic mode
        fx = self.grads_fx is not None
        _lib.check(self.lib.snerf_nerfacto_head_input_bwd(self._p(b["ghx"]), self._p(self.cams), S2, R, self._p(b["gh"]),
                                                          None if fx else self._p(self.gviews["field.appearance"]),
                                                          self._p(self.fxviews["field.appearance"]) if fx else None, self._st), "head_input_bwd")
        self._mlp_bwd(self.decode, self.gviews["field.decode"], b["feat"], self.enc.output_dim, N2, b["gh"], 16, 0, b["gdens"][2], b["gfeat"],
                      self.enc.output_dim)
        if self._tiled is not None and self._in_train_step:
            # owner-computes form: bin on this stream (it reads the ray buffers), then scatter + Adam of the table as ONE pass where the sweep would go
            if self._bin_done is not None:  # binned beside the forward (early_bin): the tile pass's stream waits for it below
                torch.cuda.current_stream().wait_event(self._bin_done)
                self._bin_done = None
            else:
                with self._span("tgrid_bin.field"):
                    self._tiled.bin(self._coords[2], t, S2, b["gfeat"], self._st)
            self._tiled.coarse_levels(self._coords[2], t, S2, b["gfeat"], self.gviews["field.table"], self._st)
            if cfg.temporal_tv_weight > 0 and not early:
                self._tv_sign(0)  # the fused pass adds the TV step itself: its per-row signs must exist first (row draw order unchanged: field first)
            self._field_table_fused_adam(early)
        else:
            self._tgrid_bwd(self.enc, self._coords[2], t, S2, N2, b["gfeat"], self.gviews["field.table"])
            if early:
                if self._side is None:
                    self._side = side_stream(self.dev, "adam")
                self._field_table_sweep_async()
        # proposal supervision (interlevel loss, losses.py:106-121)
        for lvl in range(2):
            Sp, Np = self.S[lvl], R * self.S[lvl]
            _lib.check(self.lib.snerf_interlevel(self._p(b["sb"][2]), self._p(b["w"][2]), S2, self._p(b["sb"][lvl]), self._p(b["w"][lvl]), Sp, R,
                                                 cfg.interlevel_loss_mult / (R * S2), self._p(b["inter_rays"][lvl]),
                                                 self._p(b["gw"][lvl]) if proposal_grads else None, self._st), "interlevel")
            if proposal_grads:
                enc, net = self.prop_enc[lvl], self.prop_mlp[lvl]
                _lib.check(self.lib.snerf_weights_bwd(self._p(b["dens"][lvl]), self._p(b["eb"][lvl]), self._p(b["gw"][lvl]), R, Sp, self._p(b["gdens"][lvl]),
                                                      0, None, self._st), "weights_bwd")
                self._mlp_bwd(net, self.gviews[f"prop{lvl}.mlp"], b["pfeat"][lvl], enc.output_dim, Np, None, 1, 0, b["gdens"][lvl], b["gpfeat"][lvl],
                              enc.output_dim)
                self._tgrid_bwd(enc, self._coords[lvl], t, Sp, Np, b["gpfeat"][lvl], self.gviews[f"prop{lvl}.table"])
        # temporal TV of the three tables (temporal_grid.py:352-376; nerfplayer_nerfacto.py:311-316): one pass over two columns per table
        # gives the value and the per-row signed step; the gradient itself is added inside the Adam sweep (optimizer_step)
        if cfg.temporal_tv_weight > 0:
            for k in range(3):
                if early:  # all three passes were issued at the start of the backward (_field_tv_early)
                    continue
                if not (self._tiled is not None and self._in_train_step and k == 0):  # the same order of row draws either way: field, proposal 0, proposal 1
                    self._tv_sign(k)
            if early:
                # loss_dict reads the TV values and optimizer_step the proposal tables' per-row signs on the caller's stream
                torch.cuda.current_stream().wait_event(self._tv0_done)
                torch.cuda.current_stream().wait_event(self._tv12_done)

    def materialize_tv_gradient(self):
        """Adds the temporal-TV gradient into self.grads explicitly (what the Adam sweep otherwise does on the fly); for parity tests."""
        if self.cfg.temporal_tv_weight <= 0:
            return
        for k, name in enumerate(("field.table", "prop0.table", "prop1.table")):
            ca, cb = self._tv_cols[k]
            self.gviews[name][:, ca] += self._srow[k]
            self.gviews[name][:, cb] -= self._srow[k]

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        b, cfg, R = self.buf, self.cfg, self.R
        d = {"rgb_loss": b["sqerr"].sum() / (3 * R),
             "interlevel_loss": (b["inter_rays"][0].sum() + b["inter_rays"][1].sum()) / (R * self.S[2]) * cfg.interlevel_loss_mult,
             "distortion_loss": b["dist_rays"].mean() * cfg.distortion_loss_mult}
        if cfg.temporal_tv_weight > 0:
            rows = [e.embeddings.shape[0] for e in (self.enc, self.prop_enc[0], self.prop_enc[1])]
            d["temporal_tv_loss"] = sum(b["tv"][k, :, 0].sum() / rows[k] for k in range(3)) * cfg.temporal_tv_weight
        return d

    def optimizer_step(self):
        """Adam (lr * cosine schedule, eps 1e-12) over the flat buffer, gradient cleared in the sweep; the three tables go through
        snerf_adam_step_tv, which adds the temporal-TV gradient of their two columns on the fly."""
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
        off = {name: (o, n) for name, _, _, o, n in self.segments}
        swept = self._field_swept  # async_field_sweep: the field table has been stepped (and its cells converted) on the side stream already
        fo, fn = off["field.table"]
        if self.grads_fx is not None:  # fixed-point cells -> float gradients (cells cleared)
            if swept:
                ops.fx_to_float(self.grads_fx[:fo], self.grads[:fo], accumulate=True)
                ops.fx_to_float(self.grads_fx[fo + fn:], self.grads[fo + fn:], accumulate=True)
            else:
                ops.fx_to_float(self.grads_fx, self.grads, accumulate=True)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # one plain sweep per run of segments between (and after) the tables; with the TV term each table gets its own sweep that adds
        # the TV gradient of its two columns.  Every parameter is stepped exactly once (`done` = first float not yet swept).
        done = 0
        plain = lambda lo, hi: ops.adam_step(self.params[lo:hi], self.grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.step + 1, lr,
                                             eps=self.adam_eps, zero_grad=True)
        tables = ("prop0.table", "prop1.table", "field.table") if self.cfg.temporal_tv_weight > 0 else (("field.table",) if swept else ())
        for name in sorted(tables, key=lambda nm: off[nm][0]):
            o, n = off[name]
            if o > done:  # the small segments in front of this table
                plain(done, o)
            if not (swept and name == "field.table"):
                self._sweep_table(name, lr, st)
            done = o + n
        if done < self.n_params:
            plain(done, self.n_params)
        self._field_swept = False
        self.step += 1

    def random_draws(self) -> Dict[str, torch.Tensor]:
        R = self.R
        flat = torch.rand(R * 6, device=self.dev)  # single jitter: one draw per ray and level (nerfplayer_nerfacto.py:99) + background
        return {"t_rand": flat[:R].view(R, 1), "u": [flat[R:2 * R].view(R, 1), flat[2 * R:3 * R].view(R, 1)], "bg": flat[3 * R:].view(R, 3)}

    def train_step(self, rays: Dict[str, torch.Tensor], cams: torch.Tensor, target: torch.Tensor, rng: Optional[Dict[str, torch.Tensor]] = None):
        """One optimiser step; returns the rendered colours of the batch (a work buffer, valid until the next call).  With fused_ray_loss (default) the nerf
        level's weights, compositing, MSE backward, distortion term and weights' backward are one launch and the expected DEPTH of the batch is not rendered
        (no loss of this model reads it: buf["depth"] keeps the last forward(training=False)'s values); forward() + backward() called separately render it."""
        cfg = self.cfg
        anneal = anneal_value(self.step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope) \
            if cfg.use_proposal_weight_anneal else 1.0
        sstep = max(self.step - 1, 0)  # the sampler's counter is set by the AFTER_TRAIN_ITERATION callback (nerfacto.py:249-263)
        sched = float(np.clip(np.interp(sstep, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every))
        updated = self._steps_since_update > sched or sstep < 10
        rng = rng if rng is not None else self.random_draws()
        self._in_train_step = True
        try:
            out = self.forward(rays, cams, rng, anneal)
            self.backward(target, rng, proposal_grads=updated)
        finally:
            self._in_train_step = False
        self.optimizer_step()
        if updated:
            self._steps_since_update = 0
        self._steps_since_update += 1
        return out
