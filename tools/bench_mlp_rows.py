"""Backward of the tiny MLPs alone, at the K-Planes preset's shapes, through the C ABI: the default kernel for the shape (64-wide nets:
csrc/mlp_rows.hip, wave owns rows) against the workgroup-tile kernel (snerf_mlp_bwd_tile = round 4).  HIP events, 30 launches.  Dev tool.

    python tools/bench_mlp_rows.py [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soccernerfs_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    L = _lib.lib()
    out = {}
    # name, d_in, hidden, n_hidden, d_out, out_act, ldx, ldgy, ldgx, aux, N
    shapes = [("color_net 15->64->64->3", 15, 64, 2, 3, 1, 16, 3, 16, False, 4096 * 64),
              ("proposal 8->64->1 (level 0)", 8, 64, 1, 1, 0, 8, 1, 8, True, 4096 * 256),
              ("proposal 8->64->1 (level 1)", 8, 64, 1, 1, 0, 8, 1, 8, True, 4096 * 128),
              ("nerfplayer mlp_base 32->64->16", 32, 64, 1, 16, 0, 32, 16, 32, True, 4096 * 48)]
    for name, d_in, hidden, nh, d_out, out_act, ldx, ldgy, ldgx, aux, N in shapes:
        d = _lib.MlpDesc()
        d.d_in, d.d_out, d.hidden, d.n_hidden, d.hidden_act, d.out_act, d.operands = d_in, d_out, hidden, nh, 1, out_act, 1
        W = ((torch.rand(L.snerf_mlp_param_count(C.byref(d)), device=DEV) - 0.5) * 0.4)
        X = torch.rand(N, ldx, device=DEV) - 0.3
        gY = None if (aux and d_out == 1) else torch.rand(N, ldgy, device=DEV) - 0.5
        gaux = torch.rand(N, device=DEV) - 0.5 if aux else None
        gX = torch.zeros(N, ldgx, device=DEV)
        gW = torch.zeros_like(W)
        row = {}
        for label, fn in (("rows", L.snerf_mlp_bwd), ("tile", L.snerf_mlp_bwd_tile)):
            call = lambda: _lib.check(fn(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gY) if gY is not None else None, ldgy,
                                         0 if aux else -1, ops._ptr(gaux) if gaux is not None else None, ops._ptr(gX), ldgx, ops._ptr(gW), ops._stream()), label)
            row[label + "_ms"] = round(timed(call), 4)
        # weight gradients through the 16-replica workspace (snerf_mlp_bwd_ws) + the reduce that folds it into gW
        L.snerf_mlp_gw_workspace_floats.restype = C.c_int64
        ws = torch.zeros(int(L.snerf_mlp_gw_workspace_floats(C.byref(d))), device=DEV)
        row["rows_ws_ms"] = round(timed(lambda: _lib.check(L.snerf_mlp_bwd_ws(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gY) if gY is not None else None, ldgy,
                                                           0 if aux else -1, ops._ptr(gaux) if gaux is not None else None, ops._ptr(gX), ldgx, ops._ptr(ws), ops._stream()), "ws")), 4)
        row["gw_reduce_ms"] = round(timed(lambda: _lib.check(L.snerf_mlp_gw_reduce(C.byref(d), ops._ptr(ws), ops._ptr(gW), ops._stream()), "reduce")), 4)
        # where the fixed cost sits: the same launch without weight gradients (no flush) and on 2048 samples (launch + staging + flush only)
        row["rows_no_gW_ms"] = round(timed(lambda: _lib.check(L.snerf_mlp_bwd(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(N), ops._ptr(gY) if gY is not None else None, ldgy,
                                                              0 if aux else -1, ops._ptr(gaux) if gaux is not None else None, ops._ptr(gX), ldgx, None, ops._stream()), "nogw")), 4)
        for label, fn in (("rows", L.snerf_mlp_bwd), ("tile", L.snerf_mlp_bwd_tile)):
            row[label + "_2048_samples_ms"] = round(timed(lambda: _lib.check(fn(C.byref(d), ops._ptr(W), ops._ptr(X), ldx, C.c_int64(2048), ops._ptr(gY) if gY is not None else None, ldgy,
                                                                         0 if aux else -1, ops._ptr(gaux) if gaux is not None else None, ops._ptr(gX), ldgx, ops._ptr(gW), ops._stream()), "small")), 4)
        hbm = N * 4 * (ldx + (ldgy if gY is not None else 0) + (1 if aux else 0) + d_in)
        row["hbm_bytes"] = hbm
        row["rows_hbm_frac_of_8TBs"] = round(hbm / (row["rows_ms"] * 1e-3) / 8e12, 3)
        out[name] = row
        print(name, row)
    # sigma_net from the 16-bit feature tile, with the quotient epilogue, as the trainer calls it: rows kernel (csrc/mlp_rows128.hip) vs tile kernel
    for name, d_in in (("sigma_net 160->128->16 (config 2)", 160), ("sigma_net 192->128->16 (config 3)", 192)):
        N = 4096 * 64
        d = _lib.MlpDesc()
        d.d_in, d.d_out, d.hidden, d.n_hidden, d.hidden_act, d.out_act, d.operands = d_in, 16, 128, 1, 1, 0, 1
        W = ((torch.rand(L.snerf_mlp_param_count(C.byref(d)), device=DEV) - 0.5) * 0.2)
        X16 = (torch.rand(N, d_in, device=DEV) - 0.3).to(torch.bfloat16)
        gY = torch.rand(N, 16, device=DEV) - 0.5
        gaux = torch.rand(N, device=DEV) - 0.5
        G = torch.zeros(N, d_in, device=DEV)
        gW = torch.zeros_like(W)
        ws = torch.zeros(int(L.snerf_mlp_gw_workspace_floats(C.byref(d))), device=DEV)
        fl = torch.zeros(2 * N, dtype=torch.int32, device=DEV)
        cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
        row = {}
        for label, env in (("rows", None), ("tile", "0")):
            if env is not None:
                os.environ["SNERF_MLP_SIGMA_ROWS"] = env
            try:
                row[label + "_quotient_ws_ms"] = round(timed(lambda: _lib.check(L.snerf_mlp_bwd_x16_quotient_ws(
                    C.byref(d), ops._ptr(W), ops._ptr(X16), d_in, C.c_int64(N), ops._ptr(gY), 16, 15, ops._ptr(gaux), ops._ptr(G), d_in, ops._ptr(fl), N,
                    ops._ptr(cnt[0:1]), ops._ptr(cnt[1:2]), ops._ptr(ws), ops._stream()), label)), 4)
                row[label + "_quotient_ms"] = round(timed(lambda: _lib.check(L.snerf_mlp_bwd_x16_quotient(
                    C.byref(d), ops._ptr(W), ops._ptr(X16), d_in, C.c_int64(N), ops._ptr(gY), 16, 15, ops._ptr(gaux), ops._ptr(G), d_in, ops._ptr(fl), N,
                    ops._ptr(cnt[0:1]), ops._ptr(cnt[1:2]), ops._ptr(gW), ops._stream()), label)), 4)
            finally:
                os.environ.pop("SNERF_MLP_SIGMA_ROWS", None)
        hbm = N * (2 * d_in + 64 + 4 + 4 * d_in)
        row["hbm_bytes"] = hbm
        row["rows_ws_hbm_frac_of_8TBs"] = round(hbm / (row["rows_quotient_ws_ms"] * 1e-3) / 8e12, 3)
        out[name] = row
        print(name, row)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
