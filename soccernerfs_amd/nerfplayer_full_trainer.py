"""Fused training step of the reference's full NeRFPlayer model (`nerfplayer` preset, NS/configs/method_configs.py:562-614) on libsnerf:
one flat parameter / gradient / Adam buffer, preallocated work buffers, hand-derived backward, no autograd graph, no host
synchronisation inside a step.  ~80 kernel launches per step instead of the ~285 the nerfstudio-shaped autograd model issues
(soccernerfs_amd.nerfplayer.NerfplayerModel, which is pinned against the reference's own model by golden G13 and is the checker of this
file: tests/test_gpu_nerfplayer_full_trainer.py compares outputs, losses and every gradient tensor on identical draws).

Per step (NS/models/nerfplayer.py:218-343, NS/fields/nerfplayer_field.py:330-414, NS/models/nerfacto.py:235-264):

  collider (AABB) -> piecewise sampler (single jitter) -> [temporal hash grid -> 10->16->1 MLP -> trunc_exp -> weights -> PDF] x 2
  -> positions p -> deformation MLP 3->128x3->3 -> static hash grid at p AND at p + delta (one launch over 2N points)
  -> [grid features | t] -> 33->64->32 MLP (both halves) -> newness temporal grid, decomposition temporal grid -> 32->64->3 MLP
  -> softmax mixing (csrc/nerfplayer.hip) -> 32->64->64->16 MLP (density = trunc_exp(col 0)) -> 15->64x3->3 sigmoid MLP -> weights
  -> rgb / accumulation / expected depth / rendered probabilities
  -> MSE + interlevel + 1e-3 distortion + temporal TV / 4 of four tables + 0.1 (0.01 mean p_deform + mean p_new)
  -> gradients of all of it (incl. the hash grid's coordinate gradient into the deformation MLP)
  -> Adam (lr 1e-2 x cosine, eps 1e-6) with the TV gradient folded into the tables' sweeps, gradients cleared in the sweep."""
import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, ops
from .streams import side_stream
from .nerfplayer import NerfplayerModelConfig
from .tcnn_compat import Encoding, Network
from .temporal_grid import TemporalGridEncoder
from .trainer import anneal_value, cosine_lr_factor

_ACT = {"none": 0, "relu": 1, "sigmoid": 2}


def _align4(n: int) -> int:
    return (n + 3) // 4 * 4


class NerfplayerFullTrainer:
    def __init__(self, cfg: NerfplayerModelConfig, num_rays: int, aabb_scale: float = 1.0, device="cuda:0", lr: float = 1e-2,
                 adam_eps: float = 1e-6, warm_up_end: int = 512, max_steps: int = 30000, seed: int = 0, deterministic: bool = False,
                 async_table_sweeps: bool = False, mlp_operands: str = "fp32", tiled_table_backward: bool = False, tiled_hash_backward: bool = False):
        """mlp_operands: "fp32" (exact: every net on the fp32 matrix instructions -- the parity path, what G13 / G13b pin) or "bf16" (round 5: bf16 MFMA
        operands with fp32 accumulation in every net -- the fused kernels for the shapes in their table, csrc/dense_lp.hip's single layers for the
        deformation net, the 33 -> 64 -> 32 MLP and the colour head; the reference itself runs all of them in tcnn's fp16).
        async_table_sweeps (round 5): inside train_step the optimiser sweeps of the newness and decomposition tables (most of the parameters) go to a side
        stream as soon as their gradients are complete (right behind their temporal-grid backward, early in the backward pass) and are joined in front of
        the next forward's first read of those tables: they run beside the rest of the backward (hash-grid scatter, deformation net, proposal networks)
        and the next step's proposal levels.  Same arithmetic (same bits in deterministic mode).  Readers of those tables outside forward() call
        wait_params() / synchronize() first; off by default for that reason.
        tiled_table_backward (round 6): inside train_step the newness and decomposition tables' gradient scatter, temporal-TV step and Adam sweep are ONE
        owner-computes pass each (temporal_grid.TiledTableBackward, csrc/tgrid_tiles.hip): no float atomics on the deformed positions (tgrid_bwd_runs_kernel
        <false> ran at the atomic rate, 2.0 ms per step) and no dense gradient for the two tables (24 instead of 32 B per parameter).  Same mathematics, float
        sums in another order.  Needs temporal_tv_weight > 0 (the preset); not with `deterministic`; backward() outside train_step keeps the atomic scatter.
        deterministic: every gradient accumulated across samples -- the temporal-grid and hash-grid table scatters, the hash grid's coordinate
        gradient (one add per level and sample), the weight gradients of all seven nets -- goes into 2^50-scaled 64-bit cells
        (snerf_*_bwd_fx; integer adds are associative) and is converted once per step: two runs from the same state and draws give the same bits.
        Costs 8 bytes per parameter (3.7 GB at the preset) and 64-bit atomics."""
        if not cfg.disable_scene_contraction or cfg.use_same_proposal_network or cfg.num_proposal_iterations != 2 or not cfg.disable_viewing_dependent:
            raise NotImplementedError("NerfplayerFullTrainer covers the `nerfplayer` preset (AABB collider, two proposal networks, no view dependence)")
        self.cfg, self.R, self.dev = cfg, num_rays, torch.device(device)
        self.lr, self.adam_eps, self.warm_up_end, self.max_steps = lr, adam_eps, warm_up_end, max_steps
        a = aabb_scale
        self.aabb = [[-a, -a, -a], [a, a, a]]
        torch.manual_seed(seed)
        if mlp_operands not in ("fp32", "bf16"):
            raise ValueError(f"mlp_operands must be 'fp32' or 'bf16', got {mlp_operands!r}")
        self.mlp_operands = mlp_operands
        self._dense_operands = {"fp32": 0, "bf16": 1}[mlp_operands]
        fc = {"otype": "FullyFusedMLP", "activation": "ReLU"}

        def net(din, dout, h, nh, act):
            cfg_ = {**fc, "output_activation": act, "n_neurons": h, "n_hidden_layers": nh}
            d = _lib.MlpDesc()
            d.d_in, d.d_out, d.hidden, d.n_hidden, d.hidden_act, d.out_act, d.operands = din, dout, h, nh, 1, int(act == "Sigmoid"), self._dense_operands
            fused16 = self._dense_operands != 0 and bool(_lib.lib().snerf_mlp_supported(C.byref(d)))
            return Network(din, dout, cfg_, operands=mlp_operands if fused16 else "fp32")  # layer-chained nets pick their kernels in _dense_chain_*
        F = cfg.num_levels * cfg.features_per_level
        self.F = F
        # ---- modules exactly as the fields build them (nerfplayer_nerfacto_field.py:83-104; nerfplayer_field.py:223-316) ----
        self.prop_enc: List[TemporalGridEncoder] = []
        self.prop_mlp: List[Network] = []
        for args in cfg.proposal_net_args_list[:2]:
            L, H = args.get("num_levels", 8), args.get("hidden_dim", 64)
            growth = float(np.exp((np.log(args.get("max_res", 1024)) - np.log(16)) / (L - 1)))
            self.prop_enc.append(TemporalGridEncoder(input_dim=3, temporal_dim=args.get("temporal_dim", 64), num_levels=L, level_dim=2,
                                                     per_level_scale=growth, base_resolution=16, log2_hashmap_size=args.get("log2_hashmap_size", 18)))
            self.prop_mlp.append(net(2 * L, 1, H, 1, "None"))
        self.deform = net(3, 3, 128, 3, "None")                                                       # :231
        self.hash = Encoding(3, {"otype": "HashGrid", "n_levels": cfg.num_levels, "n_features_per_level": cfg.features_per_level,
                                 "log2_hashmap_size": cfg.log2_hashmap_size, "base_resolution": 16, "per_level_scale": 1.4472692012786865})  # :243
        self.stat_mlp = net(F + 1, F, 64, 1, "None")                                                  # :254
        grid = dict(input_dim=3, temporal_dim=cfg.temporal_dim, num_levels=cfg.num_levels, level_dim=cfg.features_per_level, base_resolution=16,
                    log2_hashmap_size=cfg.log2_hashmap_size, desired_resolution=1024 * 2.0 * a)
        self.newness = TemporalGridEncoder(**grid)                                                    # :270
        self.decomp = TemporalGridEncoder(**grid)                                                     # :280
        self.decomp_mlp = net(F, 3, 64, 1, "None")                                                    # :290
        self.decode = net(F, 16, 64, 2, "None")                                                       # :301
        self.head = net(15, 3, 64, 3, "Sigmoid")                                                      # :309
        for m in (self.decomp_mlp, self.decode, self.prop_mlp[0], self.prop_mlp[1]):
            assert m.fused, "these nets run through the fused MLP kernels"
        # ---- one flat buffer: [proposal_networks | fields] ----
        self.segments = []
        off = 0
        order = [(f"prop{i}.{k}", mod, attr) for i in range(2) for k, mod, attr in (("table", self.prop_enc[i], "embeddings"), ("mlp", self.prop_mlp[i], "params"))]
        self.n_proposal_segments = len(order)
        order += [("field.deform", self.deform, "params"), ("field.hash", self.hash, "params"), ("field.stat_mlp", self.stat_mlp, "params"),
                  ("field.newness", self.newness, "embeddings"), ("field.decomp", self.decomp, "embeddings"), ("field.decomp_mlp", self.decomp_mlp, "params"),
                  ("field.decode", self.decode, "params"), ("field.head", self.head, "params")]
        for name, mod, attr in order:
            n = getattr(mod, attr).numel()
            self.segments.append((name, mod, attr, off, n))
            off += _align4(n)
        self.n_params = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.grads_fx = torch.zeros(off, dtype=torch.int64, device=self.dev) if deterministic else None
        self.views, self.gviews = {}, {}
        for name, mod, attr, o, n in self.segments:
            p = getattr(mod, attr)
            self.params[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            self.views[name] = p.data
            self.gviews[name] = self.grads[o:o + n].view(p.shape)
        # ---- work buffers ----
        R = num_rays
        S0, S1 = cfg.num_proposal_samples_per_ray
        S2 = cfg.num_nerf_samples_per_ray
        self.S = (S0, S1, S2)
        N = R * S2
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.dev)
        self.buf = {
            "sb": [f(R, s + 1) for s in self.S], "eb": [f(R, s + 1) for s in self.S],
            "dens": [f(R, s) for s in self.S], "w": [f(R, s) for s in self.S], "gw": [f(R, s) for s in self.S], "gdens": [f(R, s) for s in self.S],
            "pfeat": [f(R * S0, self.prop_enc[0].output_dim), f(R * S1, self.prop_enc[1].output_dim)],
            "pout": [f(R * S0, 1), f(R * S1, 1)],
            "gpfeat": [f(R * S0, self.prop_enc[0].output_dim), f(R * S1, self.prop_enc[1].output_dim)],
            "mid": f(R, S2), "tN": f(N), "t2N": f(2 * N),
            "x2": f(2 * N, 3),                               # rows [0,N) = p, rows [N,2N) = p + deformation(p)
            "dh": [f(N, 128) for _ in range(3)], "delta": f(N, 3), "gdh": [f(N, 128), f(N, 128)],
            "enc2": f(2 * N, F), "genc2": f(2 * N, F), "gx2": z(2 * N, 3),
            "sx": z(2 * N, 36), "gsx": f(2 * N, 36),         # [grid features | t | pad], row stride 36 (16-B aligned rows)
            "sh": f(2 * N, 64), "gsh": f(2 * N, 64), "sv": f(2 * N, F), "gsv": f(2 * N, F),
            "vnew": f(N, F), "gvnew": f(N, F), "dfeat": f(N, F), "gdfeat": f(N, F),
            "logits": f(N, 3), "glogits": f(N, 3), "probs": f(N, 3), "gprobs": f(N, 3), "v": f(N, F), "gv": f(N, F),
            "h": f(N, 16), "gh": z(N, 16), "hh": [f(N, 64) for _ in range(3)], "ghh": [f(N, 64), f(N, 64)],
            "rgb": f(N, 3), "grgb": f(N, 3), "tmpN": f(N),
            "rgb_out": f(R, 3), "acc": f(R), "depth": f(R), "sqerr": z(R), "dist_rays": f(R), "inter_rays": [f(R), f(R)],
            "tv": z(4, 64, 16),
        }
        self._gx_fx = torch.zeros(N * 3, dtype=torch.int64, device=self.dev) if deterministic else None  # the deformed half's coordinate gradient
        self._encs = [self.newness, self.decomp, self.prop_enc[0], self.prop_enc[1]]  # temporal-TV order of nerfplayer.py:329-333
        self._enc_names = ["field.newness", "field.decomp", "prop0.table", "prop1.table"]
        self._srow = [z(e.embeddings.shape[0]) for e in self._encs]
        amin = torch.tensor(self.aabb[0], dtype=torch.float32, device=self.dev)
        self._amin, self._arange = amin, torch.tensor(self.aabb[1], dtype=torch.float32, device=self.dev) - amin
        self._cvec = torch.tensor([0.0, 0.01, 1.0], dtype=torch.float32, device=self.dev)  # prob loss weights (nerfplayer.py:339-341)
        self.lib = _lib.lib()
        self.step = 0
        self._steps_since_update = 0
        self.tv_rows: Optional[List[int]] = None  # parity hook: fixed table rows [newness, decomp, prop0, prop1]
        self._tv_cols = [(0, 1)] * 4
        self.launches = 0  # libsnerf launches of the last step (diagnostics)
        self.async_table_sweeps = bool(async_table_sweeps)
        self._tiled, self._tiled_hash, self._hash_swept, self._bin_done, self._hash_bin_done, self.early_bin = None, None, False, None, None, True
        if tiled_table_backward and not deterministic and cfg.temporal_tv_weight > 0:
            from .temporal_grid import TiledTableBackward

            self._tiled = [TiledTableBackward(self.newness, N, first_tiled_level=0), TiledTableBackward(self.decomp, N, first_tiled_level=0)]
            self._tiled[1].share_bins_of(self._tiled[0])  # both grids are evaluated at the same (undeformed) positions and times: one binning pass (early_bin)
            if tiled_hash_backward:
                # the static hash grid's table the same way (csrc/hashgrid_tiles.hip); opt-in: see DESIGN section 7 for where it pays
                from .tcnn_compat import TiledHashTableBackward

                self._tiled_hash = TiledHashTableBackward(self.hash, 2 * N)  # both halves of x2 (undeformed and deformed positions) in one pass
        self._side, self._sweeps_done, self._swept, self._in_train_step, self._tv01_done = None, None, (), False, None

    # ---- helpers ----
    def _p(self, t, off_floats: int = 0):
        return C.c_void_p(t.data_ptr() + 4 * off_floats)

    def _ck(self, rc, what):
        self.launches += 1
        _lib.check(rc, what)

    def wait_params(self):
        """The current stream waits for the asynchronous table sweeps (no host block)."""
        # the event is KEPT until the next sweep replaces it: waiting for a completed event is free, and a later reader on ANOTHER stream (a checkpoint
        # save, a side-stream evaluation) that calls wait_params() is then ordered behind the sweep too
        if self._sweeps_done is not None:
            torch.cuda.current_stream().wait_event(self._sweeps_done)

    def synchronize(self):
        self.wait_params()
        torch.cuda.synchronize()

    def _tv_sign(self, k: int):
        """Value + per-row signed step of table k's temporal TV (order of nerfplayer.py:329-333); the row draw is the reference's randint."""
        enc = self._encs[k]
        row = self.tv_rows[k] if self.tv_rows is not None else int(torch.randint(0, len(enc._index_list_host), [1]).item())
        ca, cb = enc._index_list_host[row]
        self._tv_cols[k] = (ca, cb)
        rows_, gc = enc.embeddings.shape
        self._ck(self.lib.snerf_tgrid_tv_sign(self._p(enc.embeddings), C.c_int64(rows_), gc, ca, cb, float(self.cfg.temporal_tv_weight) / 4.0,
                                              self._p(self.buf["tv"][k]), 64, self._p(self._srow[k]), self._st), "tv_sign")

    def _sweep_table(self, k: int, lr: float, st):
        """Adam over temporal table k on the current stream (st = its handle) with the TV gradient of its two columns added on the fly."""
        o, n = next((o, n) for name, _, _, o, n in self.segments if name == self._enc_names[k])
        enc = self._encs[k]
        ca, cb = self._tv_cols[k]
        rows_, gc = enc.embeddings.shape
        sl = slice(o, o + n)
        self._ck(self.lib.snerf_adam_step_tv(self._p(self.params[sl]), self._p(self.grads[sl]), self._p(self.exp_avg[sl]), self._p(self.exp_avg_sq[sl]),
                                             C.c_int64(rows_), gc, ca, cb, self._p(self._srow[k]), lr, 0.9, 0.999, self.adam_eps, self.step + 1, 1.0, 1,
                                             None, st), "adam_step_tv")

    def _early_tv(self):
        """async_table_sweeps: the TV passes of the newness (k = 0) and decomposition (k = 1) tables read parameters only -- on the side stream at the START of
        the backward instead of on the caller's stream (the critical path of this model) between the tables' gradient scatters and their sweeps."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = side_stream(self.dev, "adam")  # the process-wide sweep stream (streams.py)
        self._side.wait_stream(main)
        keep = self._st
        with torch.cuda.stream(self._side):
            self._st = C.c_void_p(self._side.cuda_stream)
            try:
                for k in (0, 1):
                    self._tv_sign(k)
            finally:
                self._st = keep
            self._tv01_done = self._side.record_event()

    def _early_table_sweeps(self):
        """async_table_sweeps: TV pass + Adam sweep of the newness (k = 0) and decomposition (k = 1) tables on the side stream, behind everything the
        caller's stream holds (their gradient scatters)."""
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)  # (the two tables' TV passes have been on this stream since the start of the backward: _early_tv)
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
        with torch.cuda.stream(self._side):
            st = C.c_void_p(self._side.cuda_stream)
            for k in (0, 1):
                if self.grads_fx is not None:
                    gv = self.gviews[self._enc_names[k]]
                    o = (gv.data_ptr() - self.grads.data_ptr()) // 4
                    ops.fx_to_float(self.grads_fx[o:o + gv.numel()], gv.view(-1), accumulate=True)
                self._sweep_table(k, lr, st)
            self._sweeps_done = self._side.record_event()
        self._swept = (0, 1)

    def _tiled_fused_adam(self, gouts, on_side: bool):
        """tiled_table_backward: scatter + temporal TV + Adam of the newness (k = 0) and decomposition (k = 1) tables, one owner-computes pass each, over the
        tiles filed by `bin`; on the side stream (async_table_sweeps) it reads only the tilers' own buffers and the two feature-gradient buffers, which the
        next step rewrites after wait_params() at the earliest."""
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)

        def run(st):
            for k in (0, 1):
                o, n = next((o, n) for name, _, _, o, n in self.segments if name == self._enc_names[k])
                sl = slice(o, o + n)
                self._tiled[k].scatter_adam(gouts[k], None, self.params[sl], self.exp_avg[sl], self.exp_avg_sq[sl], lr, self.step + 1, self.adam_eps,
                                            tv_cols=self._tv_cols[k], srow=self._srow[k], stream=st)
                self.launches += 1

        if on_side:
            self._side.wait_stream(torch.cuda.current_stream())  # (the two tables' TV passes have been on this stream since the start of the backward)
            with torch.cuda.stream(self._side):
                run(C.c_void_p(self._side.cuda_stream))
                self._sweeps_done = self._side.record_event()
        else:
            run(self._st)
        self._swept = (0, 1)

    def _tgrid_fwd(self, enc, co, times, spr, N, out):
        self._ck(self.lib.snerf_tgrid_encode_fwd(C.byref(enc.desc), self._p(enc.embeddings), C.byref(co), None, self._p(times), spr, C.c_int64(N), self._p(out),
                                                 self._st), "tgrid_fwd")

    def _pfx(self, gview: torch.Tensor, off_cells: int = 0):
        """The fixed-point cells behind a view of self.grads (deterministic mode), as a pointer."""
        o = (gview.data_ptr() - self.grads.data_ptr()) // 4 + off_cells
        return C.c_void_p(self.grads_fx.data_ptr() + 8 * o)

    def gradients_to_float(self):
        """Deterministic mode: fold the fixed-point cells into self.grads (cells cleared).  optimizer_step does this itself; callers that read
        self.gviews after backward() call it first."""
        if self.grads_fx is None:
            return
        lo = 0
        for o, n in sorted((o, n) for name, _, _, o, n in self.segments if name in [self._enc_names[k] for k in self._swept]):
            if o > lo:  # the tables swept on the side stream have been converted there (and are being written by it)
                ops.fx_to_float(self.grads_fx[lo:o], self.grads[lo:o], accumulate=True)
            lo = o + n
        if lo < self.n_params:
            ops.fx_to_float(self.grads_fx[lo:], self.grads[lo:], accumulate=True)

    def _tgrid_bwd(self, enc, co, times, spr, N, gout, gtable):
        if self.grads_fx is not None:
            self._ck(self.lib.snerf_tgrid_encode_bwd_fx(C.byref(enc.desc), C.byref(co), None, self._p(times), spr, C.c_int64(N), self._p(gout),
                                                        self._pfx(gtable), self._st), "tgrid_bwd_fx")
            return
        self._ck(self.lib.snerf_tgrid_encode_bwd(C.byref(enc.desc), C.byref(co), None, self._p(times), spr, C.c_int64(N), self._p(gout), self._p(gtable),
                                                 self._st), "tgrid_bwd")

    def _mlp_fwd(self, net, X, ldx, N, Y, ldy, aux_col=-1, aux=None):
        self._ck(self.lib.snerf_mlp_fwd(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(Y), ldy, aux_col,
                                        self._p(aux) if aux is not None else None, self._st), "mlp_fwd")

    def _mlp_bwd(self, net, gW, X, ldx, N, gY, ldgy, aux_col, gaux, gX, ldgx):
        if self.grads_fx is not None:
            self._ck(self.lib.snerf_mlp_bwd_fx(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(gY) if gY is not None else None,
                                               ldgy, aux_col, self._p(gaux) if gaux is not None else None, self._p(gX) if gX is not None else None, ldgx,
                                               self._pfx(gW), self._st), "mlp_bwd_fx")
            return
        self._ck(self.lib.snerf_mlp_bwd(C.byref(net.desc), self._p(net.params), self._p(X), ldx, C.c_int64(N), self._p(gY) if gY is not None else None,
                                        ldgy, aux_col, self._p(gaux) if gaux is not None else None, self._p(gX) if gX is not None else None, ldgx,
                                        self._p(gW), self._st), "mlp_bwd")

    def _dense_chain_fwd(self, net, acts, X, ldx, x_off, N, outs):
        """Bias-free dense layers of `net` chained from snerf_dense_fwd; outs[l] = layer l's (activated) output, contiguous [N, dims[l+1]]."""
        woff, cur, ld, xo = 0, X, ldx, x_off
        for l, y in enumerate(outs):
            K, M = net.dims[l], net.dims[l + 1]
            if self._dense_operands:
                self._ck(self.lib.snerf_dense_fwd_lp(self._p(net.params, woff), K, M, _ACT[acts[l]], self._p(cur, xo), ld, C.c_int64(N), self._p(y), y.stride(0),
                                                     self._dense_operands, self._st), "dense_fwd_lp")
            else:
                self._ck(self.lib.snerf_dense_fwd(self._p(net.params, woff), K, M, _ACT[acts[l]], self._p(cur, xo), ld, C.c_int64(N), self._p(y), y.stride(0),
                                                  self._st), "dense_fwd")
            woff += K * M
            cur, ld, xo = y, y.stride(0), 0

    def _dense_chain_bwd(self, net, gname, acts, X, ldx, x_off, N, outs, gY, ldgy, scratch, gX, ldgx, gx_off=0):
        """Backward of _dense_chain_fwd: gY = gradient of the last output; scratch = two [N, width] buffers for the hidden gradients;
        gX (may be None) receives the gradient of the chain's input."""
        woffs = [0]
        for l in range(len(outs)):
            woffs.append(woffs[-1] + net.dims[l] * net.dims[l + 1])
        g, ldg = gY, ldgy
        gW = self.gviews[gname]
        for l in reversed(range(len(outs))):
            K, M = net.dims[l], net.dims[l + 1]
            xin, ldi, xo = (X, ldx, x_off) if l == 0 else (outs[l - 1], outs[l - 1].stride(0), 0)
            if l == 0:
                gx, ldgx_, go = gX, ldgx, gx_off
            else:
                gx, ldgx_, go = scratch[l % 2], scratch[l % 2].stride(0), 0
            fx = self.grads_fx is not None
            if self._dense_operands:
                self._ck(self.lib.snerf_dense_bwd_lp(
                    self._p(net.params, woffs[l]), K, M, _ACT[acts[l]], self._p(xin, xo), ldi, C.c_int64(N), self._p(outs[l]), outs[l].stride(0), self._p(g), ldg,
                    self._p(gx, go) if gx is not None else None, ldgx_, None if fx else self._p(gW, woffs[l]), self._pfx(gW, woffs[l]) if fx else None,
                    self._dense_operands, self._st), "dense_bwd_lp")
                g, ldg = gx, ldgx_
                continue
            self._ck((self.lib.snerf_dense_bwd_fx if fx else self.lib.snerf_dense_bwd)(
                self._p(net.params, woffs[l]), K, M, _ACT[acts[l]], self._p(xin, xo), ldi, C.c_int64(N), self._p(outs[l]), outs[l].stride(0), self._p(g), ldg,
                self._p(gx, go) if gx is not None else None, ldgx_, self._pfx(gW, woffs[l]) if fx else self._p(gW, woffs[l]), self._st), "dense_bwd")
            g, ldg = gx, ldgx_

    def _resample(self, lvl, rand, anneal):
        b, a = self.buf, _lib.ResampleArgs()
        a.density, a.ebins_prev, a.weights_out = b["dens"][lvl].data_ptr(), b["eb"][lvl].data_ptr(), b["w"][lvl].data_ptr()
        a.sbins_prev, a.nears, a.fars = b["sb"][lvl].data_ptr(), self.rays["nears"].data_ptr(), self.rays["fars"].data_ptr()
        if rand is None:
            a.u_mode = 2
        else:
            a.u_mode, a.u_or_rand, a.rand_cols = 1, rand.data_ptr(), rand.shape[-1]
        a.sbins_out, a.ebins_out = b["sb"][lvl + 1].data_ptr(), b["eb"][lvl + 1].data_ptr()
        a.R, a.S_prev, a.S, a.kind = self.R, self.S[lvl], self.S[lvl + 1], 1  # UniformLinDispPiecewise spacing (ray_samplers.py:242-243)
        a.anneal, a.histogram_padding, a.eps = anneal, 0.01, 1e-5
        self._ck(self.lib.snerf_pdf_resample(C.byref(a), self._st), "pdf_resample")

    # ---- forward ----
    def forward(self, rays: Dict[str, torch.Tensor], rng: Dict[str, torch.Tensor], anneal: float, training: bool = True):
        """rays: origins [R,3], directions [R,3] (unit), times [R,1]; rng: t_rand [R,1], u [2 x [R,1]], bg [R,3] (training) -- eval renders on
        the white background without jitter (nerfplayer.py:228-231).  Returns rgb [R,3] (a work buffer)."""
        cfg, b, R, F = self.cfg, self.buf, self.R, self.F
        self._st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.launches = 0
        o, d = ops._f32c(rays["origins"], "origins"), ops._f32c(rays["directions"], "directions")
        t = ops._f32c(rays["times"], "times").reshape(-1)
        rays = dict(rays)
        rays["nears"], rays["fars"] = ops.aabb_collide(o, d, self.aabb, 0.0, training)  # AABBBoxCollider(scene_box): near_plane 0
        self.rays = rays
        t_rand = rng["t_rand"] if training else None
        self._ck(self.lib.snerf_spaced_bins(self._p(rays["nears"]), self._p(rays["fars"]), self._p(t_rand) if t_rand is not None else None,
                                            t_rand.shape[-1] if t_rand is not None else 0, R, self.S[0], 1, self._p(b["sb"][0]), self._p(b["eb"][0]),
                                            self._st), "spaced_bins")
        self._coords = []
        for lvl in range(2):
            co = ops.coords_from_rays(o, d, t, b["eb"][lvl], self.aabb, False)
            self._coords.append(co)
            S, N = self.S[lvl], R * self.S[lvl]
            enc, net = self.prop_enc[lvl], self.prop_mlp[lvl]
            self._tgrid_fwd(enc, co, t, S, N, b["pfeat"][lvl])
            self._mlp_fwd(net, b["pfeat"][lvl], enc.output_dim, N, b["pout"][lvl], 1, 0, b["dens"][lvl])
            self._resample(lvl, rng["u"][lvl] if training else None, anneal)
        # ---- main field on the N = R * S2 final samples ----
        S, N = self.S[2], R * self.S[2]
        eb = b["eb"][2]
        torch.add(eb[:, :-1], eb[:, 1:], out=b["mid"])
        b["mid"].div_(2)                                                    # (starts + ends) / 2 (rays.py:54)
        p = b["x2"][:N].view(R, S, 3)
        torch.mul(d[:, None, :], b["mid"][:, :, None], out=p)
        p.add_(o[:, None, :]).sub_(self._amin).div_(self._arange)           # SceneBox.get_normalized_positions (scene_box.py:55-65)
        b["tN"].view(R, S).copy_(t[:, None].expand(R, S))
        self._dense_chain_fwd(self.deform, ("relu", "relu", "relu", "none"), b["x2"], 3, 0, N, b["dh"] + [b["delta"]])
        torch.add(b["x2"][:N], b["delta"], out=b["x2"][N:])                 # deformed = x + deformation_field(x) (:346)
        self._ck(self.lib.snerf_hashgrid_encode_fwd(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"]), C.c_int64(2 * N), self._p(b["enc2"]),
                                                    self._st), "hashgrid_fwd")
        b["sx"][:, :F].copy_(b["enc2"])
        b["sx"][:N, F].copy_(b["tN"])
        b["sx"][N:, F].copy_(b["tN"])                                       # cat([stationary_field(.), t]) (:349-351)
        self._dense_chain_fwd(self.stat_mlp, ("relu", "none"), b["sx"], 36, 0, 2 * N, [b["sh"], b["sv"]])
        self._pts = ops.coords_from_points(b["x2"])                         # explicit points for the temporal grids (pts [N,3], mode 0)
        self.wait_params()  # the newness / decomposition tables' sweeps of the last step (async_table_sweeps)
        if self._tiled is not None and self._in_train_step and training and self.early_bin:
            # the binning passes of the tiled backward need positions only: on a side stream NOW, beside the rest of the forward and the backward's MLP chain,
            # instead of on the critical chain in front of the tile passes.  ONE pass for the newness and decomposition grids (same positions, same times).
            main = torch.cuda.current_stream()
            sb = side_stream(self.dev, "sort")
            sb.wait_stream(main)
            with torch.cuda.stream(sb):
                stb = C.c_void_p(sb.cuda_stream)
                self._tiled[0].bin(self._pts, b["tN"], 1, None, stb)
                self._bin_done = sb.record_event()
                if self._tiled_hash is not None:
                    self._tiled_hash.bin(b["x2"], None, stb)
                    self._hash_bin_done = sb.record_event()
            self.launches += 5 + (5 if self._tiled_hash is not None else 0)
        self._tgrid_fwd(self.newness, self._pts, b["tN"], 1, N, b["vnew"])
        self._tgrid_fwd(self.decomp, self._pts, b["tN"], 1, N, b["dfeat"])
        self._mlp_fwd(self.decomp_mlp, b["dfeat"], F, N, b["logits"], 3)
        self._ck(self.lib.snerf_nerfplayer_mix_fwd(self._p(b["logits"]), self._p(b["sv"]), self._p(b["sv"], N * F), self._p(b["vnew"]), C.c_int64(N), F,
                                                   self._p(b["probs"]), self._p(b["v"]), self._st), "mix_fwd")
        self._mlp_fwd(self.decode, b["v"], F, N, b["h"], 16, 0, b["dens"][2])   # density = trunc_exp(column 0) (:374-377)
        self._dense_chain_fwd(self.head, ("relu", "relu", "relu", "sigmoid"), b["h"], 16, 1, N, b["hh"] + [b["rgb"]])  # geo features = columns 1..15
        self._ck(self.lib.snerf_weights_fwd(self._p(b["dens"][2]), self._p(eb), R, S, self._p(b["w"][2]), self._st), "weights_fwd")
        a = _lib.RenderArgs()
        a.weights, a.rgb, a.ebins = b["w"][2].data_ptr(), b["rgb"].data_ptr(), eb.data_ptr()
        if training:
            a.bg_mode, a.bg = 0, rng["bg"].data_ptr()
        else:
            self._white = torch.ones(3, dtype=torch.float32, device=self.dev)
            a.bg_mode, a.bg = 2, self._white.data_ptr()
        a.R, a.S, a.training = R, S, int(training)
        a.rgb_out, a.acc_out, a.depth_expected = b["rgb_out"].data_ptr(), b["acc"].data_ptr(), b["depth"].data_ptr()
        self._ck(self.lib.snerf_render_fwd(C.byref(a), self._st), "render_fwd")
        return b["rgb_out"]

    def rendered_probs(self) -> torch.Tensor:
        """DecompositionRenderer (renderers.py:422-444): sum_s weights * probs -> [R,3]."""
        R, S = self.R, self.S[2]
        return (self.buf["w"][2][:, :, None] * self.buf["probs"].view(R, S, 3)).sum(1)

    # ---- backward ----
    def backward(self, target: torch.Tensor, rng: Dict[str, torch.Tensor], proposal_grads: bool):
        cfg, b, R, F = self.cfg, self.buf, self.R, self.F
        S2, N = self.S[2], R * self.S[2]
        t = self.rays["times"].reshape(-1)
        target = ops._f32c(target, "target")
        # a backward that raised after its asynchronous sweeps were issued never reached optimizer_step(): start from clean flags, or this step's TV pass
        # and sweep of those tables would be skipped
        self._swept, self._tv01_done, self._hash_swept = (), None, False
        early = bool(self.async_table_sweeps and self._in_train_step and cfg.temporal_tv_weight > 0)
        if cfg.temporal_tv_weight > 0:
            b["tv"].zero_()
        if early:
            self._early_tv()
        self._ck(self.lib.snerf_render_mse_bwd(self._p(b["w"][2]), self._p(b["rgb"]), self._p(rng["bg"]), 0, self._p(b["rgb_out"]), self._p(target),
                                               2.0 / (3 * R), R, S2, self._p(b["gw"][2]), self._p(b["grgb"]), self._p(b["sqerr"]), self._st), "render_mse_bwd")
        self._ck(self.lib.snerf_distortion(self._p(b["w"][2]), self._p(b["sb"][2]), R, S2, cfg.distortion_loss_mult / R, self._p(b["dist_rays"]),
                                           self._p(b["gw"][2]), 1, self._st), "distortion")
        # probability regulariser: mult * (0.01 mean_r P_deform + mean_r P_new), P = sum_s w * probs (nerfplayer.py:336-341)
        k = cfg.prob_reg_loss_mult / R
        torch.mul(b["w"][2].view(N, 1), self._cvec, out=b["gprobs"])
        b["gprobs"].mul_(k)                                                  # d loss / d probs
        torch.mul(b["probs"][:, 1], 0.01, out=b["tmpN"])                     # probs . (0, 0.01, 1) without a BLAS call (a [N,3] x [3] gemv took 0.14 ms)
        b["tmpN"].add_(b["probs"][:, 2])
        b["gw"][2].view(-1).add_(b["tmpN"], alpha=k)                          # d loss / d weights
        self._ck(self.lib.snerf_weights_bwd(self._p(b["dens"][2]), self._p(b["eb"][2]), self._p(b["gw"][2]), R, S2, self._p(b["gdens"][2]), 0, None,
                                            self._st), "weights_bwd")
        # colour head -> geometry features (columns 1..15 of gh; column 0 = density enters the decode net through gaux)
        self._dense_chain_bwd(self.head, "field.head", ("relu", "relu", "relu", "sigmoid"), b["h"], 16, 1, N, b["hh"] + [b["rgb"]], b["grgb"], 3, b["ghh"],
                              b["gh"], 16, 1)
        self._mlp_bwd(self.decode, self.gviews["field.decode"], b["v"], F, N, b["gh"], 16, 0, b["gdens"][2], b["gv"], F)
        self._ck(self.lib.snerf_nerfplayer_mix_bwd(self._p(b["probs"]), self._p(b["sv"]), self._p(b["sv"], N * F), self._p(b["vnew"]), self._p(b["gv"]),
                                                   self._p(b["gprobs"]), C.c_int64(N), F, self._p(b["gsv"]), self._p(b["gsv"], N * F), self._p(b["gvnew"]),
                                                   self._p(b["glogits"]), self._st), "mix_bwd")
        self._mlp_bwd(self.decomp_mlp, self.gviews["field.decomp_mlp"], b["dfeat"], F, N, b["glogits"], 3, -1, None, b["gdfeat"], F)
        if self._tiled is not None and self._in_train_step:
            # owner-computes form: bin both tables' touches here (the pass reads the deformed positions), then scatter + TV + Adam per table as one pass
            gouts = (b["gvnew"], b["gdfeat"])
            if self._bin_done is not None:  # binned beside the forward (early_bin, one pass for both tables)
                torch.cuda.current_stream().wait_event(self._bin_done)
                self._bin_done = None
            else:
                # late binning without the zero-gradient filter as well: the two tables share ONE set of records (share_bins_of)
                self._tiled[0].bin(self._pts, b["tN"], 1, None, self._st)
                self.launches += 5
            if not early:
                for k in (0, 1):
                    self._tv_sign(k)  # the fused pass adds the TV step itself (row draw order unchanged: newness, decomposition first)
            self._tiled_fused_adam(gouts, early)
        else:
            self._tgrid_bwd(self.decomp, self._pts, b["tN"], 1, N, b["gdfeat"], self.gviews["field.decomp"])
            self._tgrid_bwd(self.newness, self._pts, b["tN"], 1, N, b["gvnew"], self.gviews["field.newness"])
            if early:
                self._early_table_sweeps()
        self._dense_chain_bwd(self.stat_mlp, "field.stat_mlp", ("relu", "none"), b["sx"], 36, 0, 2 * N, [b["sh"], b["sv"]], b["gsv"], F, [b["gsh"], b["gsh"]],
                              b["gsx"], 36)
        b["genc2"].copy_(b["gsx"][:, :F])
        # static hash grid: table gradient from both halves, coordinate gradient only for the deformed half (x itself carries no gradient)
        if self._tiled_hash is not None and self._in_train_step:
            # owner-computes form (round 6): one binning pass over all 2N points, then scatter + Adam of the table as one pass; the coordinate gradient of
            # the deformed half is a gather (the atomic kernel without a table gradient)
            th = self._tiled_hash
            lc = th.plan.first_tiled_level
            gth = self.gviews["field.hash"]
            if self._hash_bin_done is not None:
                torch.cuda.current_stream().wait_event(self._hash_bin_done)
                self._hash_bin_done = None
            else:
                th.bin(b["x2"], b["genc2"], self._st)
            b["gx2"][N:].zero_()
            # the coarse levels (every point of the batch in a handful of tiles) through the atomic kernel: table gradient from both halves ...
            th.coarse_levels(b["x2"], b["genc2"], gth, self._st)
            # ... and the coordinate gradient of the deformed half: levels [0, lc) were handled just above for the table, so one gather-only launch per range
            self._ck(self.lib.snerf_hashgrid_encode_bwd(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"], 3 * N), C.c_int64(N),
                                                        self._p(b["genc2"], N * F), None, self._p(b["gx2"], 3 * N), self._st), "hashgrid_bwd (coordinates)")
            lr_h = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
            oh, nh = next((o, n) for name, _, _, o, n in self.segments if name == "field.hash")
            th.scatter_adam(b["x2"], b["genc2"], gth if lc > 0 else None, self.params[oh:oh + nh], self.exp_avg[oh:oh + nh], self.exp_avg_sq[oh:oh + nh], lr_h,
                            self.step + 1, self.adam_eps, stream=self._st)
            self.launches += 6
            self._hash_swept = True
        elif self.grads_fx is not None:
            gt = self._pfx(self.gviews["field.hash"])
            self._ck(self.lib.snerf_hashgrid_encode_bwd_fx(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"]), C.c_int64(N), self._p(b["genc2"]),
                                                           gt, None, self._st), "hashgrid_bwd_fx")
            self._ck(self.lib.snerf_hashgrid_encode_bwd_fx(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"], 3 * N), C.c_int64(N),
                                                           self._p(b["genc2"], N * F), gt, self._p(self._gx_fx), self._st), "hashgrid_bwd_fx")
            ops.fx_to_float(self._gx_fx, b["gx2"][N:].view(-1))  # written, cells cleared
        else:
            self._ck(self.lib.snerf_hashgrid_encode_bwd(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"]), C.c_int64(N), self._p(b["genc2"]),
                                                        self._p(self.gviews["field.hash"]), None, self._st), "hashgrid_bwd")
            b["gx2"][N:].zero_()
            self._ck(self.lib.snerf_hashgrid_encode_bwd(C.byref(self.hash.desc), self._p(self.hash.params), self._p(b["x2"], 3 * N), C.c_int64(N),
                                                        self._p(b["genc2"], N * F), self._p(self.gviews["field.hash"]), self._p(b["gx2"], 3 * N), self._st),
                     "hashgrid_bwd")
        self._dense_chain_bwd(self.deform, "field.deform", ("relu", "relu", "relu", "none"), b["x2"], 3, 0, N, b["dh"] + [b["delta"]], b["gx2"][N:], 3,
                              b["gdh"], None, 0)
        # proposal supervision (interlevel loss, losses.py:106-121)
        for lvl in range(2):
            Sp, Np = self.S[lvl], R * self.S[lvl]
            self._ck(self.lib.snerf_interlevel(self._p(b["sb"][2]), self._p(b["w"][2]), S2, self._p(b["sb"][lvl]), self._p(b["w"][lvl]), Sp, R,
                                               cfg.interlevel_loss_mult / (R * S2), self._p(b["inter_rays"][lvl]),
                                               self._p(b["gw"][lvl]) if proposal_grads else None, self._st), "interlevel")
            if proposal_grads:
                enc, net = self.prop_enc[lvl], self.prop_mlp[lvl]
                self._ck(self.lib.snerf_weights_bwd(self._p(b["dens"][lvl]), self._p(b["eb"][lvl]), self._p(b["gw"][lvl]), R, Sp, self._p(b["gdens"][lvl]),
                                                    0, None, self._st), "weights_bwd")
                self._mlp_bwd(net, self.gviews[f"prop{lvl}.mlp"], b["pfeat"][lvl], enc.output_dim, Np, None, 1, 0, b["gdens"][lvl], b["gpfeat"][lvl],
                              enc.output_dim)
                self._tgrid_bwd(enc, self._coords[lvl], t, Sp, Np, b["gpfeat"][lvl], self.gviews[f"prop{lvl}.table"])
        # temporal TV of the four tables, weight / 4 (nerfplayer.py:329-333): values + per-row signed steps; the gradient is added in the Adam sweep
        if cfg.temporal_tv_weight > 0:
            for k_ in range(4):
                if k_ not in self._swept:  # the same order of row draws either way
                    self._tv_sign(k_)
            if early:
                torch.cuda.current_stream().wait_event(self._tv01_done)  # loss_dict reads those tables' TV values on the caller's stream

    def materialize_tv_gradient(self):
        """Adds the temporal-TV gradient into self.grads explicitly (what the Adam sweep otherwise does on the fly); for parity tests."""
        if self.cfg.temporal_tv_weight <= 0:
            return
        for k, name in enumerate(self._enc_names):
            ca, cb = self._tv_cols[k]
            self.gviews[name][:, ca] += self._srow[k]
            self.gviews[name][:, cb] -= self._srow[k]

    def loss_dict(self) -> Dict[str, torch.Tensor]:
        b, cfg, R = self.buf, self.cfg, self.R
        d = {"rgb_loss": b["sqerr"].sum() / (3 * R),
             "interlevel_loss": (b["inter_rays"][0].sum() + b["inter_rays"][1].sum()) / (R * self.S[2]) * cfg.interlevel_loss_mult,
             "distortion_loss": b["dist_rays"].mean() * cfg.distortion_loss_mult}
        if cfg.temporal_tv_weight > 0:
            d["temporal_tv_loss"] = sum(b["tv"][k, :, 0].sum() / e.embeddings.shape[0] for k, e in enumerate(self._encs)) * cfg.temporal_tv_weight / 4.0
        pm = self.rendered_probs().mean(0)
        d["prob_loss"] = (0.01 * pm[1] + pm[2]) * cfg.prob_reg_loss_mult
        return d

    def optimizer_step(self):
        """Adam (lr x cosine schedule) over the flat buffer, gradient cleared in the sweep; the four temporal tables go through
        snerf_adam_step_tv, which adds the temporal-TV gradient of their two columns on the fly.  Every float is swept exactly once."""
        lr = self.lr * cosine_lr_factor(self.step, self.warm_up_end, self.max_steps, 0.0)
        self.gradients_to_float()
        off = {name: (o, n) for name, _, _, o, n in self.segments}
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ho, hn = off["field.hash"]
        hash_done = self._hash_swept  # tiled_table_backward: the hash table has been stepped by its owner-computes pass already

        def plain(lo, hi):
            """the plain sweep over [lo, hi) -- around the hash table when that has been stepped already"""
            parts = [(lo, hi)]
            if hash_done and lo < ho + hn and hi > ho:
                parts = [(lo, ho), (ho + _align4(hn), hi)]
            for a_, b_ in parts:
                if b_ > a_:
                    ops.adam_step(self.params[a_:b_], self.grads[a_:b_], self.exp_avg[a_:b_], self.exp_avg_sq[a_:b_], self.step + 1, lr, eps=self.adam_eps, zero_grad=True)
                    self.launches += 1

        done = 0
        if self.cfg.temporal_tv_weight > 0:
            for k in sorted(range(4), key=lambda i: off[self._enc_names[i]][0]):
                o, n = off[self._enc_names[k]]
                if o > done:
                    plain(done, o)
                if k not in self._swept:
                    self._sweep_table(k, lr, st)
                done = o + n
        if done < self.n_params:
            plain(done, self.n_params)
        self._hash_swept = False
        self._swept = ()
        self.step += 1

    def random_draws(self) -> Dict[str, torch.Tensor]:
        R = self.R
        flat = torch.rand(R * 6, device=self.dev)  # single jitter: one draw per ray and level + background
        return {"t_rand": flat[:R].view(R, 1), "u": [flat[R:2 * R].view(R, 1), flat[2 * R:3 * R].view(R, 1)], "bg": flat[3 * R:].view(R, 3)}

    def train_step(self, rays: Dict[str, torch.Tensor], target: torch.Tensor, rng: Optional[Dict[str, torch.Tensor]] = None):
        cfg = self.cfg
        anneal = anneal_value(self.step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope) \
            if cfg.use_proposal_weight_anneal else 1.0
        sstep = max(self.step - 1, 0)  # the sampler's counter is set by the AFTER_TRAIN_ITERATION callback (nerfacto.py:249-263)
        sched = float(np.clip(np.interp(sstep, [0, cfg.proposal_warmup], [0, cfg.proposal_update_every]), 1, cfg.proposal_update_every))
        updated = self._steps_since_update > sched or sstep < 10
        rng = rng if rng is not None else self.random_draws()
        self._in_train_step = True
        try:
            out = self.forward(rays, rng, anneal)
            self.backward(target, rng, proposal_grads=updated)
        finally:
            self._in_train_step = False
        self.optimizer_step()
        if updated:
            self._steps_since_update = 0
        self._steps_since_update += 1
        return out
