#!/usr/bin/env python3
"""bench.py -- train rays/s of the MI355X-native K-Planes hot path (BASELINE.json metric).

One "step" = one full training iteration of the `k-planes` preset (BASELINE.json configs[1]) on a synthetic
Broadcast-style batch of 4096 rays per GPU: pixel sampling + image gather + ray generation + AABB collider +
3-level proposal sampling + multiscale K-Planes gather + MLPs + compositing + all losses + backward +
plane regularisers + (N>1: one RCCL all-reduce of the flat gradient buffer) + Adam.  Inputs (image cache,
camera tables, parameters) are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel
(HIP events around its launches inside the timed region) and `cpu_baseline` (the CPU oracle -- the checker, a
"port" of the reference's pure-PyTorch path -- timed on the host cores on a bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured achievable copy rate
DEFAULT_OPERANDS = "bf16"  # BASELINE.json configs[1]: "1xMI355X bf16"
PROFILE_TAG = "r06"        # profiles/<tag>_pmc_traffic.json, profiles/<tag>_psnr_*.json are quoted (with their source) in the JSON line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: the launcher's WORLD_SIZE when one started this script, else 1")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step (k-planes preset: 4096)")
    ap.add_argument("--images", type=int, default=0, help="override the number of synthetic training images (default 19 cams x 33 frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-standin", action="store_true", help="skip the stock-PyTorch-ROCm stand-in of the reference rate (N = 1 only; ~10 s)")
    ap.add_argument("--breakdown", action="store_true", help="print a per-kernel time table to stderr after the timed run")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream backward (debug / A-B)")
    ap.add_argument("--start-step", type=int, default=0, help="pretend this many optimiser steps are done: >= 5000 gives the steady-state "
                    "schedule (proposal networks updated every 5th step, kplanes.py:254-259) instead of the every-step schedule of early training")
    ap.add_argument("--sync-adam", action="store_true", help="A-B: field-plane optimiser sweep on the main stream instead of its own stream under the next step's proposal levels")
    ap.add_argument("--grad-transport", default="fp32", choices=["fp32", "bf16"], help="world > 1, sharded optimiser: element type of the field-plane "
                    "gradient on the links.  fp32 (default: the headline `value` is the reference's DDP arithmetic, NS/pipelines/base_pipeline.py:244-246); "
                    "bf16 = half the reduce-scatter bytes (single-GPU emulation over 10 paired 30 k-step seeds: PSNR -0.03 +- 0.12 dB, "
                    "profiles/r03_psnr_30k_bf16_emulated_bf16_transports.json) -- timed in the same process as the labelled leg `bf16_transports`")
    ap.add_argument("--param-transport", default="fp32", choices=["fp32", "bf16"], help="world > 1, sharded optimiser: fp32 (default) = all-gather the new "
                    "field planes (reference arithmetic); bf16 = all-gather the parameter UPDATES in bf16 and apply them identically on every rank")
    ap.add_argument("--no-bf16-transport-leg", action="store_true", help="world > 1: skip the extra timed region with both transports in bf16")
    ap.add_argument("--cabi-allreduce", action="store_true", help="world > 1 with --no-shard: the flat gradient all-reduce goes through libsnerf's own "
                    "RCCL communicator (snerf_allreduce_grads) instead of torch.distributed")
    ap.add_argument("--no-shard", action="store_true", help="world > 1: one all-reduce of the whole gradient + replicated Adam (A-B)")
    ap.add_argument("--mlp-operands", default=DEFAULT_OPERANDS, choices=["fp32", "bf16", "fp16"], help="MFMA operand type of sigma_net, color_net and the proposal "
                    "nets: bf16 = bf16 operands with fp32 accumulation (BASELINE config 2 names bf16; tcnn computes these nets in fp16); fp32 = exact (the parity path)")
    ap.add_argument("--trained-until", type=int, default=5000, help="third leg: really train to this step (untimed, ~12 s), then time K steady-state steps (0 or --no-steady-state: skip)")
    ap.add_argument("--time-sorted-rays", action="store_true", help="A/B: every batch in order of frame time (ops.sort_rays_by_time); faster field forward, but 1.5 %% slower trained steps: profiles/r03_kernels.md section 11")
    ap.add_argument("--no-fused-field", action="store_true", help="A/B: unfused forward (gather, sigma_net, color_net as three kernels); default is the fused forward kernel (csrc/field_fused.hip) with the unfused backward")
    ap.add_argument("--pipeline-sweep", default="off", choices=["coarse_first", "fine_first", "off"],
                    help="pass B and the optimiser sweep of the field planes pipelined by scale (KPlanesTrainConfig.pipeline_sweep); off: one pass B, then one sweep (A/B)")
    ap.add_argument("--pipeline-sweep-config3", default="off", choices=["coarse_first", "fine_first", "off"],
                    help="the same for the config-3 leg (measured: 4.57 / 4.63 ms pipelined, 4.60 / 4.64 ms not)")
    ap.add_argument("--sort-before-field-fwd", action="store_true", help="A/B: the nerf level's sample sort before the wait for the sweep instead of after the field forward (measured slower: trainer.py)")
    ap.add_argument("--pass-b-main-stream", action="store_true", help="A/B: pass B of the field scatter on the caller's stream (round 3) instead of on the sweep's stream beside the next step's head")
    ap.add_argument("--no-fused-proposal", action="store_true", help="A/B: proposal levels as gather + net kernels instead of the fused density kernel (bit-identical densities)")
    ap.add_argument("--no-quotient-epilogue", action="store_true", help="A/B: round 3's flow -- G = gfeat .* feat from a separate pass over fp32 features instead of the sigma_net backward's epilogue")
    ap.add_argument("--no-quotient-scatter", action="store_true", help="A/B: product form of the field's sorted scatter (gradvec + 1 GB of per-plane gradient vectors) instead of the quotient form")
    ap.add_argument("--no-alone", action="store_true", help="skip the 20 extra steps that time the optimiser sweep alone as ONE launch (profiling passes: every step of the run "
                    "then launches the sweep the same way, so rocprofv3's per-kernel averages are means over like launches)")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the second timed region (steady-state schedule + IST importance sampler)")
    ap.add_argument("--cpu-steps", type=int, default=4, help="oracle train steps timed for cpu_baseline")
    ap.add_argument("--no-fused-ray-loss", action="store_true", help="A-B: the nerf level's weights / render / MSE / distortion / weights-backward as five kernels instead of one")
    ap.add_argument("--interleave-prop-levels", action="store_true", help="A/B: proposal backward interleaved by stage (both net backwards before the two plane scatters) instead of level by level")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 leg (K-Planes multiscale 1-32, IST range 0.75, fps-downsample 4; N = 1 only, ~15 s)")
    ap.add_argument("--no-config4", action="store_true", help="skip the config-4 leg (nerfplayer-nerfacto preset on the synthetic stadium-players scene; N = 1 only, ~20 s)")
    ap.add_argument("--no-tiled-config4", action="store_true", help="A/B: config 4 with the round-5 form (run-length atomic scatter + dense Adam sweep of the main table) instead of "
                    "the owner-computes pass (csrc/tgrid_tiles.hip)")
    ap.add_argument("--leg-steps", type=int, default=20, help="timed steps of the config-3 / config-4 legs")
    return ap.parse_args()


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; falls back to the logical count."""
    try:
        seen, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                seen.add((phys, ln.split(":", 1)[1].strip()))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_all_cores_leg(timeout_s=75):
    """BASELINE.md section 3 asks for torch.set_num_threads(all physical cores).  The step is ~1500 small ops, so that many threads only
    synchronise; to put a NUMBER on it without risking the bench's run time, config 1 is timed with every physical core in a child process
    under a time limit (no GPU is touched there)."""
    import subprocess

    n = _physical_cores()
    code = ("import sys, json, torch; sys.path.insert(0, %r); torch.set_num_threads(%d); torch.manual_seed(0);"
            "from oracle import torch_standin as TS; r = TS.time_train_steps('cpu', 256, steps=3, warmup=1, model=TS.CONFIG1);"
            "print(json.dumps({'rays_per_s': r['rays_per_s'], 'ms_per_step': r['seconds_per_step'] * 1e3}))") % (ROOT, n)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return {"threads": n, "value": r["rays_per_s"], "unit": "rays/s", "ms_per_step": r["ms_per_step"], "sample": "config 1, 3 full train steps after 1 warm-up"}
    except subprocess.TimeoutExpired:
        return {"threads": n, "value": None, "note": f"did not finish 4 train steps of config 1 within {timeout_s} s (the 16-thread run takes ~0.1 s per step)"}
    except Exception as e:
        return {"threads": n, "value": None, "note": f"{type(e).__name__}: {e}"[:200]}


def cpu_baseline(rays_per_step=256, steps=4):
    """The stock-PyTorch restatement of the reference's K-Planes train step (oracle/torch_standin.py: F.grid_sample per plane, Linear
    stacks, autograd, torch.optim.Adam) on the host cores: a 256-ray slice of config 2 (the preset's planes) and config 1 itself
    (BASELINE.md section 3)."""
    from oracle import torch_standin as TS  # baseline / checker code; only timed here, never on the product path

    torch.manual_seed(0)
    # many small ops: beyond ~16 intra-op threads torch only adds synchronisation overhead (256 threads: 2000x slower)
    n_threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(n_threads)
    c2 = TS.time_train_steps("cpu", rays_per_step, steps=steps, warmup=1, model=TS.PRESET)
    c1 = TS.time_train_steps("cpu", 256, steps=20, warmup=2, model=TS.CONFIG1)
    return {"value": c2["rays_per_s"], "unit": "rays/s", "cores": n_threads, "kind": "port",
            "sample": f"{steps} full train steps (fwd + autograd bwd + torch.optim.Adam, k-planes preset planes = config 2, fp32) of {rays_per_step} rays each, "
                      "stock PyTorch on the host cores (oracle/torch_standin.py)",
            "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(), "host_physical_cores": _physical_cores(), "torch_threads": n_threads,
            "torch": torch.__version__, "config1_all_physical_cores": cpu_all_cores_leg(),
            "config1": {"value": c1["rays_per_s"], "unit": "rays/s", "ms_per_step": c1["seconds_per_step"] * 1e3,
                        "sample": "BASELINE.json configs[0]: single scale (64,64,64,8), C=32, proposals (128^3,8)/(256^3,8), samples 256/128/64, 256 rays/step, "
                                  "fp32, 20 full train steps after 2 warm-up"}}


def reference_standin(dev, rays, hip_rays_per_s):
    """BASELINE.md section 2's stand-in for the unpublished reference rate: the same algorithm in stock PyTorch-ROCm ops on THIS MI355X
    (oracle/torch_standin.py), a handful of steps outside the HIP path's timed region.  fp32 and with the MLP matmuls under bf16 autocast
    (the reference trains with mixed_precision=True); the ratio uses the FASTER of the two."""
    from oracle import torch_standin as TS  # baseline code, never the product path

    runs = {}
    for name, ac in (("fp32", None), ("autocast_bf16", torch.bfloat16)):
        try:
            r = TS.time_train_steps(dev, rays, steps=8, warmup=3, model=TS.PRESET, autocast=ac)
            runs[name] = {"rays_per_s": r["rays_per_s"], "ms_per_step": r["seconds_per_step"] * 1e3, "steps": r["steps"]}
        except Exception as e:  # the stand-in must not take the bench line down
            runs[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        torch.cuda.empty_cache()
    ok = [v["rays_per_s"] for v in runs.values() if "rays_per_s" in v]
    best = max(ok) if ok else None
    return {"what": "stock PyTorch-ROCm K-Planes train step on the same MI355X: F.grid_sample per plane + Linear stacks + autograd + torch.optim.Adam, "
                    f"k-planes preset, {rays} rays/step (BASELINE.md section 2: the stand-in for the reference's unpublished CUDA rate)",
            "runs": runs, "value": best, "unit": "rays/s", "hip_over_standin": (hip_rays_per_s / best) if best else None,
            "note": "vs_baseline stays null: BASELINE.md holds no published number for this metric; this ratio is the measured stand-in for north_star's >= 10x"}


def _die(code, msg):
    print(f"bench.py: FATAL: {msg}", file=sys.stderr, flush=True)
    sys.exit(code)


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as the reference's train.py does
    (NSR/scripts/train.py:187-200 mp.spawn, :124-137 NCCL init) -- one CHILD process per GPU through torch.distributed.run, rendezvous on
    127.0.0.1.  Runs BEFORE this process touches the GPU (torch.cuda.device_count() does not initialise it on this image) and never
    exec()s: the children's stdout / stderr are ours, their exit code is ours."""
    import socket
    import subprocess

    one_device = os.environ.get("SNERF_BENCH_ONE_DEVICE") == "1"
    if not one_device and os.environ.get("SNERF_BENCH_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()
        if have < args.gpus:
            _die(4, f"--gpus {args.gpus} but only {have} HIP device(s) visible: refusing to run a smaller job under the name of a larger one")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def _merge_sweep_spans(kt, tr):
    """KPlanesTrainConfig.pipeline_sweep launches the field planes' optimiser sweep twice per step (the finest scale's planes / the coarser scales').  Their
    HIP-event spans are folded into ONE "adam_planes.field" entry = mean over all launches (what rocprofv3's per-kernel average of plane_reg_kernel<32,true>
    is) and returned separately as well, each with its own algorithmic bytes (32 B / parameter of its range)."""
    parts = {k: kt.pop(k) for k in ("adam_planes.field.fine", "adam_planes.field.coarse") if k in kt}
    if not parts:
        return None
    n = sum(v[1] for v in parts.values())
    kt["adam_planes.field"] = (sum(v[0] * v[1] for v in parts.values()) / n, n)
    cut, tot = tr._finest_offset(), tr.field_planes.numel
    nbytes = {"adam_planes.field.fine": 32 * (tot - cut), "adam_planes.field.coarse": 32 * cut}
    return {k.rsplit(".", 1)[1]: {"avg_launch_ms": round(v[0], 4), "launches_timed": v[1], "algorithmic_per_launch": nbytes[k],
                                   "frac": round(nbytes[k] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)} for k, v in parts.items()}


def _hbm_roofline(kernel, alg_bytes, ms, launches, **extra):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_per_launch": int(alg_bytes), "avg_launch_ms": ms, "launches_timed": launches, **extra}


def config3_leg(dev, args):
    """BASELINE.json configs[2]: K-Planes multiscale-res 1-32 + IST range 0.75, fps-downsample 4 (REF/README.md:39-44 changes those three
    settings of the k-planes preset only, so spacetime_resolution stays (64,64,64,100), method_configs.py:515): 578 367 744 parameters, 19
    cameras x 25 frames, 15 % IST importance rays, steady-state schedule.  Whole train steps, timed like the headline."""
    from soccernerfs_amd import ops, synthetic
    from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    R, steps = args.rays, args.leg_steps
    torch.manual_seed(20231029)
    cfg = KPlanesTrainConfig(mlp_operands=args.mlp_operands, multiscale_res=(1, 2, 4, 8, 16, 32), spacetime_resolution=(64, 64, 64, 100),
                             proposal_resolutions=((128, 128, 128, 100), (256, 256, 256, 100)),
                             pipeline_sweep=args.pipeline_sweep_config3 if args.pipeline_sweep_config3 != "off" else "")
    tr = KPlanesTrainer(cfg, R, dev)
    tr.step = 6000  # steady-state schedule, IST active (iters_to_start_ist = 2000)
    cams = synthetic.make_cameras(20, 960, 540)
    data = synthetic.render_dataset(cams, synthetic.frame_times(100, 4), list(range(19)), dev, chunk_rows=540)
    M, H, W = data["images"].shape[:3]
    ist = compute_ist(data["images"], data["cam_id"], data["times"], ist_range=0.75)
    batch = {"image": data["images"], "image_idx": torch.arange(M, device=dev), "ist_weights": ist, "iter_steps": 6000}
    sampler = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
    DynamicBasedPixelSampler.prepare(batch)

    def step():
        idx = sampler.sample_method(R, M, H, W, batch=batch, device=dev)
        target = data["images"][idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=tr.aabb, near_plane=cfg.near_plane, training=True)
        tr.train_step(rays, target)

    for _ in range(max(args.warmup, 8)):
        step()
    cand = ["adam_planes.field", "adam_planes.field.fine", "adam_planes.field.coarse", "kplanes_scatter_sorted.field", "kplanes_field_fwd", f"mlp_bwd.{32 * 6}x128x1"]
    tr.enable_kernel_timing(cand)
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    tr.synchronize()
    dt = time.perf_counter() - t0
    kt = tr.kernel_times_ms()
    tr.disable_kernel_timing()
    sweep_parts = _merge_sweep_spans(kt, tr)
    S2, ns, C = cfg.num_nerf_samples_per_ray, 6, cfg.feature_dim
    gather = R * S2 * ns * 6 * 4 * C * 4
    nl = tr.field_sweep_launches  # 2: the sweep is pipelined with pass B by scale (finest scale's planes, then the rest): mean bytes per launch
    alg = {"adam_planes.field": (32 * tr.field_planes.numel // nl, "plane_reg_kernel<32,true> (Adam + K-Planes regularisers, field planes): 32 B / parameter"
                                 + (f"; {nl} launches per step (finest scale, then the coarser ones, pipelined with pass B): mean over both" if nl > 1 else "")),
           "kplanes_scatter_sorted.field": (2 * gather, "pass B of the sorted scatter: read-modify-write of every touched texel"),
           "kplanes_field_fwd": (gather + R * S2 * (16 + 2 * C * ns + 64), "field_fwd_kernel (gather + sigma_net + color_net fused)")}
    ktl = {k: v for k, v in kt.items() if k in alg}
    dom = max(ktl, key=lambda k: ktl[k][0] * ktl[k][1])
    out = {"value": R * steps / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": args.mlp_operands,
           "config": {"workload": "K-Planes multiscale-res 1-32 (6 scales x 6 planes, C = 32, spacetime_resolution (64,64,64,100)) + IST range 0.75, fps-downsample 4 "
                                  "(19 cameras x 25 frames 960x540), 15 % importance rays, steady-state schedule (step 6000+), full train step incl. Adam",
                      "params": int(tr.n_params), "images": int(M), "rays_per_gpu": R},
           "roofline": _hbm_roofline(alg[dom][1], alg[dom][0], ktl[dom][0], ktl[dom][1],
                                     other_kernels_ms={k: round(v[0], 4) for k, v in kt.items() if k != dom},
                                     **({"launch_breakdown": sweep_parts} if sweep_parts and dom == "adam_planes.field" else {}))}
    del tr, data, ist, batch
    torch.cuda.empty_cache()
    return out


def config4_leg(dev, args):
    """BASELINE.json configs[3]: NeRFPlayer-nerfacto (hash grid + temporal decomposition; method_configs.py:616-660) on the stadium-players scene:
    30 wide-angle training cameras high in the bleachers, aabb [-1,1]^3, fps_downsample 1 (stadiumwide_dataparser.py:94-112).  Rays are CAMERA rays of
    the synthetic scene (synthetic.make_stadium_cameras / shade_stadium), uniform pixels as the preset's VanillaDataManager draws them.  A subset of
    each camera's 100 frames is rendered (the timed kernels do not see how many images back the sampler); time stamps and the appearance-embedding
    table are those of the full 3000-image set."""
    from soccernerfs_amd import ops, synthetic
    from soccernerfs_amd.nerfplayer_nerfacto import NerfplayerNerfactoModelConfig
    from soccernerfs_amd.nerfplayer_trainer import NerfplayerTrainer

    R, steps = args.rays, args.leg_steps
    torch.manual_seed(20231029)
    n_cams, n_frames, frames_rendered = 30, 100, 8
    cams = synthetic.make_stadium_cameras(n_cams, 6, 960, 540)
    frame_ids = torch.linspace(0, n_frames - 1, frames_rendered).long()
    data = synthetic.render_dataset(cams, frame_ids.float() / (n_frames - 1), list(range(n_cams)), dev, chunk_rows=540, variant="stadium")
    M, H, W = data["images"].shape[:3]
    full_index = (data["cam_id"] * n_frames + frame_ids.to(dev).repeat(n_cams)).contiguous()  # image m of the rendered subset -> its index among the 3000
    mc = NerfplayerNerfactoModelConfig()
    tiled = not args.no_tiled_config4
    tr = NerfplayerTrainer(mc, R, n_cams * n_frames, aabb_scale=1.0, device=dev, async_field_sweep=True, mlp_operands=args.mlp_operands if args.mlp_operands in ("fp32", "bf16") else "fp32",
                           tiled_field_backward=tiled)
    tr.step = 600  # past the learning-rate warm-up

    def step():
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, data["images"])
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"])
        tr.train_step(rays, full_index[idx[:, 0]].contiguous(), target)

    for _ in range(max(args.warmup, 8)):
        step()
    tr.enable_kernel_timing(["adam_tv.field.table", "tgrid_tiles_adam.field", "tgrid_bin.field", "tgrid_fwd.field", "tgrid_bwd.field", "tgrid_fwd.prop", "tgrid_bwd.prop"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = tr.kernel_times_ms()
    tr.disable_kernel_timing()
    S2 = mc.num_nerf_samples_per_ray
    L, Cl = mc.num_levels, mc.features_per_level
    n_table = tr.enc.embeddings.numel()
    # SURVEY 8d, config 4, BOTH byte conventions for the main table's gather: 8 corners x 3 live floats (two time rows of a column pair + ...) x 4 B per
    # level = 1536 B per sample algorithmic; 8 corners x one 64-B sector per level = 8192 B per sample at sector granularity
    tg_alg, tg_sector = R * S2 * L * 8 * 3 * 4, R * S2 * L * 8 * 64
    sweep = kt.get("tgrid_tiles_adam.field") if tiled else kt.get("adam_tv.field.table")
    if tiled:
        # round 6: scatter + temporal TV + Adam of the main table as ONE owner-computes pass (csrc/tgrid_tiles.hip).  Algorithmic bytes per launch: p, m, v
        # read and written once = 24 B per parameter (the dense gradient is neither read nor cleared: it does not exist for this table), plus the
        # per-step records (4 B written by the binning pass, 4 B read here) and what a record's walk reads (16 B position + time, 8 B of gfeat)
        n_rec = int(tr._tiled.tile_base[tr._tiled.plan.n_tiles])
        roof = _hbm_roofline("tt_tiles_kernel<2,1> over the main temporal grid: gradient scatter (LDS, one owner per 256-row tile) + temporal TV + Adam in one pass; "
                             "24 B / parameter (p, m, v read and written; no dense gradient) + 28 B per binned record",
                             24 * n_table + 28 * n_rec, sweep[0], sweep[1],
                             records_per_launch=n_rec,
                             note="launched on a side stream right behind the binning pass (NerfplayerTrainer async_field_sweep + tiled_field_backward): it runs beside "
                                  "the proposal networks' backward and the next step's ray generation / proposal levels, so its duration includes that sharing "
                                  "(alone: tools/bench_tgrid_tiles.py).  Replaces tgrid_bwd_runs_kernel (1.26 ms at the float-atomic rate) + adam_tv_kernel "
                                  "(32 B / parameter): bench.py --no-tiled-config4 times that form") if sweep else None
    else:
        roof = _hbm_roofline("adam_tv_kernel over the main temporal grid (Adam + temporal-TV, 32 B / parameter: p,g,m,v read, p,m,v written, g cleared)",
                             32 * n_table, sweep[0], sweep[1],
                             note="round 5: launched on a side stream right behind the table's gradient scatter (NerfplayerTrainer async_field_sweep): it runs beside the "
                                  "proposal networks' backward and the next step's ray generation / proposal levels, so its duration includes that sharing "
                                  "(alone: tools/bench_nerfplayer.py --fused --stadium --sync-sweep)") if sweep else None
    fwd = kt.get("tgrid_fwd.field")
    out = {"value": R * steps / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f32" if tr.mlp_operands == "fp32" else tr.mlp_operands,
           "dtype_note": "tables, sampling, compositing, losses, gradient accumulation and the optimiser f32; " + ("every net exact f32" if tr.mlp_operands == "fp32" else
                         "bf16 MFMA operands with f32 accumulation in the decode net and the colour head (the proposal nets' 16-wide shape has no 16-bit kernel: f32); the reference runs its nets in tcnn fp16"),
           "config": {"workload": "nerfplayer-nerfacto preset (temporal hash grid L=16 C=2 log2T=19 temporal_dim 64, proposals L=5 log2T=17, samples 256/96/48) on the synthetic "
                                  "stadium-players scene: 30 wide-angle training cameras in the bleachers, aabb [-1,1]^3, 100 frames (fps_downsample 1), camera rays, "
                                  "uniform pixels; full train step incl. Adam over every table",
                      "params": int(tr.n_params), "images_rendered": int(M), "images_in_dataset": n_cams * n_frames, "rays_per_gpu": R},
           "roofline": roof,
           "tgrid_fwd_main": None if fwd is None else {
               "avg_launch_ms": fwd[0], "launches_timed": fwd[1],
               "algorithmic": {"bytes_per_launch": tg_alg, "frac_of_hbm_peak": tg_alg / (fwd[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "convention": "SURVEY 8d: 16 levels x 8 corners x 3 floats x 4 B = 1536 B per sample"},
               "sector_granular": {"bytes_per_launch": tg_sector, "frac_of_hbm_peak": tg_sector / (fwd[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "convention": "SURVEY 8d: 16 levels x 8 corner rows x one 64-B sector = 8 KB per sample (no cache credit; above 1 means caches / "
                                                 "run-length reuse removed traffic)"}},
           "other_kernels_ms": {k: round(v[0], 4) for k, v in kt.items()}}
    del tr, data
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus is None:
        # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's rank count is the request; an EXPLICIT --gpus that
        # disagrees with the launcher stays fatal below
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a SCALE record must never carry a rank count other than the one it was asked for
        _die(2, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    # test hooks (tests/test_gpu_sharded.py, tests/test_bench_launch_cpu.py): exercise the multi-rank code path of this script on a
    # ONE-GPU box -- every rank on device 0, collectives through gloo (staged via host memory by dist.py).  Never set by the driver.
    one_device = os.environ.get("SNERF_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("SNERF_BENCH_BACKEND", "nccl")
    rank_check_only = os.environ.get("SNERF_BENCH_RANK_CHECK_ONLY") == "1"  # CPU test of the launcher: rendezvous + rank count, then stop
    if rank_check_only:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        counted = 1
        if world > 1:
            dist.init_process_group(backend="gloo")
            probe = torch.ones(1)
            if os.environ.get("SNERF_BENCH_TEST_DROP_RANK") == str(rank):  # test hook: this rank is not counted
                probe.zero_()
            dist.all_reduce(probe)
            counted = int(probe.item())
            dist.destroy_process_group()
        if counted != args.gpus:
            _die(3, f"{counted} rank(s) counted by the all-reduce, --gpus {args.gpus} asked for")
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rank_check_only": True, "ranks_counted_by_all_reduce": counted}))
        return
    if world > 1 and backend == "nccl" and not one_device and torch.cuda.device_count() < world:
        _die(4, f"{world} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the HIP path has no fallback)"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)
        pg = dist.group.WORLD

    from soccernerfs_amd import ops, synthetic
    from soccernerfs_amd.trainer import KPlanesTrainConfig, KPlanesTrainer

    # what the collectives actually run on: read back from the live communicator (a SCALE record can check N ranks against it)
    comm_info = {"backend": None, "world_size": 1, "note": "single process: no communicator"}
    if world > 1:
        import torch.distributed as dist

        probe = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(probe)  # SUM of ones over the communicator = the number of ranks that really take part
        comm_info = {"backend": dist.get_backend(pg) + (" (RCCL over xGMI)" if backend == "nccl" else ""), "world_size": dist.get_world_size(pg),
                     "ranks_counted_by_all_reduce": int(probe.item()), "devices_visible": torch.cuda.device_count()}
        if comm_info["ranks_counted_by_all_reduce"] != args.gpus:
            _die(3, f"{comm_info['ranks_counted_by_all_reduce']} rank(s) counted by the all-reduce, --gpus {args.gpus} asked for")

    # each rank draws its own rays: seed + rank (NSR/scripts/train.py:84)
    torch.manual_seed(20231029 + rank)
    cfg = KPlanesTrainConfig(mlp_operands=args.mlp_operands, fused_field=not args.no_fused_field, quotient_scatter=not args.no_quotient_scatter,
                             fused_ray_loss=not args.no_fused_ray_loss, fused_proposal=not args.no_fused_proposal,
                             quotient_epilogue=not args.no_quotient_epilogue, pass_b_beside_head=not args.pass_b_main_stream, sort_before_field_fwd=args.sort_before_field_fwd, pipeline_sweep="" if args.pipeline_sweep == "off" else args.pipeline_sweep,
                             interleave_proposal_levels=args.interleave_prop_levels)  # the k-planes preset
    R = args.rays
    trainer = KPlanesTrainer(cfg, R, dev, process_group=pg)
    trainer.overlap = not args.no_overlap
    if args.no_shard:
        trainer.shard_optimizer = False
        if args.cabi_allreduce and world > 1 and backend == "nccl":
            from soccernerfs_amd.dist import CAbiComm

            trainer.cabi_comm = CAbiComm(pg, dev)
    trainer.step = args.start_step
    trainer.async_field_adam = not args.sync_adam
    trainer.grad_transport = args.grad_transport
    trainer.param_transport = args.param_transport
    trainer.synchronize()
    init_params = trainer.params.clone()  # the trained-state leg restarts from here (0.6 GB)

    # ---- synthetic Broadcast-style data, resident in HBM ----
    cams = synthetic.make_cameras(20, 960, 540)
    times = synthetic.frame_times(100, 3)
    train_cams = list(range(19))
    if args.images:
        per_cam = max(1, args.images // len(train_cams))
        times = times[:per_cam]
    data = synthetic.render_dataset(cams, times, train_cams, dev, chunk_rows=540)
    images = data["images"]
    M, H, W = images.shape[:3]

    time_sorted = args.time_sorted_rays
    time_key, n_time_keys = ops.image_time_keys(data["times"])

    def one_step():
        # uniform pixel sampler (PixelSampler.sample_method, NS/data/pixel_samplers.py:74-77) + image gather (:111-123)
        idx, target = ops.sample_pixels_uniform(torch.rand(R, 3, device=dev), M, H, W, images)
        if time_sorted:  # the batch in order of frame time (free: a batch is a set; ops.sort_rays_by_time)
            idx, target = ops.sort_rays_by_time(idx, time_key, n_time_keys, target)
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=trainer.aabb,
                                 near_plane=cfg.near_plane, training=True)
        return trainer.train_step(rays, target)

    # steady-state region: the schedule past step 5000 (proposal networks updated every 5th step) with DynamicBasedPixelSampler active
    # (15 % of the rays drawn from the IST maps, NS/data/pixel_samplers.py:340-426).  The maps and their prefix sums are computed once,
    # outside the timed region, as the reference's CacheDataloader does at start-up (NS/data/utils/dataloaders.py:78-91).
    steady = not args.no_steady_state
    if steady:
        from soccernerfs_amd.pixel_samplers import DynamicBasedPixelSampler, compute_ist

        ist = compute_ist(images, data["cam_id"], data["times"], ist_range=1.0)  # method_configs.py:503
        batch = {"image": images, "image_idx": torch.arange(M, device=dev), "ist_weights": ist, "iter_steps": 6000}
        sampler = DynamicBasedPixelSampler(R, is_pixel_ratio=0.15, iters_to_start_ist=2000)
        DynamicBasedPixelSampler.prepare(batch)
        ist_fraction = float((ist > 0).float().mean())

    def one_step_steady():
        idx = sampler.sample_method(R, M, H, W, batch=batch, device=dev)
        if time_sorted:
            idx = ops.sort_rays_by_time(idx, time_key, n_time_keys)
        target = images[idx[:, 0], idx[:, 1], idx[:, 2]].float() / 255.0
        rays = ops.generate_rays(idx, data["fx"], data["fy"], data["cx"], data["cy"], data["c2w"], data["times"], aabb=trainer.aabb,
                                 near_plane=cfg.near_plane, training=True)
        return trainer.train_step(rays, target)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        one_step()
    # candidates for "the dominant kernel" (single kernels; HIP events around each launch inside the timed region)
    CAND = ["adam_planes.field", "adam_planes.field.fine", "adam_planes.field.coarse", "adam_step", "kplanes_scatter_sorted.field", "kplanes_gradvec.field", "kplanes_gather_bwd.field", "mlp_bwd.160x128x1",
            "kplanes_gather_fwd.field", "kplanes_gather_bwd.prop", "kplanes_field_fwd", "kplanes_quotient_prepare"]
    COMM = list(trainer.COMM_WAIT_SPANS) if world > 1 else []
    trainer.enable_kernel_timing(CAND + COMM)
    elapsed = timed(one_step, args.steps)
    kt = trainer.kernel_times_ms()
    trainer.disable_kernel_timing()
    sweep_launches = trainer.field_sweep_launches  # 2 when the sweep is pipelined with pass B by scale (KPlanesTrainConfig.pipeline_sweep)
    sweep_parts = _merge_sweep_spans(kt, trainer)

    def comm_report(kt_, n_steps):
        """world > 1: what one step puts on the links and how long the compute chain stood waiting for them (HIP events on the chain's
        own stream around every wait: see KPlanesTrainer._comm_wait)."""
        lb = trainer.link_bytes_per_step()
        waits = {k: {"ms_per_step": round(kt_[k][0] * kt_[k][1] / n_steps, 4), "waits_per_step": round(kt_[k][1] / n_steps, 2)} for k in COMM if k in kt_}
        exposed = sum(v["ms_per_step"] for v in waits.values())
        return {"link_bytes_per_step_per_gpu": {k: int(v) for k, v in lb.items()},
                "link_bytes_note": "bytes each rank sends (= receives) per step: (W-1)/W x elements x element size per reduce-scatter / all-gather, twice that per all-reduce",
                "exposed_ms_per_step": round(exposed, 4), "exposed_by_wait": waits,
                "exposed_note": "time the compute chain's stream stood at a collective's wait (events on that stream before and after the wait); the rest of "
                                "the collectives' duration ran under kernels.  With gloo (test hook) collectives block the host instead, so this reads ~0",
                "xgmi_floor_ms": round(lb["total"] / (7 * 153e9) * 1e3, 4),
                "xgmi_floor_note": "link_bytes total / (7 links x 153 GB/s): the time the step's bytes need with every xGMI link of this GPU busy in one direction"}

    comm_timed = comm_report(kt, args.steps) if world > 1 else None
    # world > 1: the same K steps once more with BOTH transports of the sharded step in bf16 -- a labelled second leg, never `value`
    bf16_leg = None
    if world > 1 and trainer._sharded() and not args.no_bf16_transport_leg and (args.grad_transport, args.param_transport) == ("fp32", "fp32"):
        trainer.synchronize()
        barrier()
        trainer.grad_transport = trainer.param_transport = "bf16"
        for _ in range(max(2, args.warmup // 4)):
            one_step()
        trainer.enable_kernel_timing(COMM)
        elb = timed(one_step, args.steps)
        comm16 = comm_report(trainer.kernel_times_ms(), args.steps)
        trainer.disable_kernel_timing()
        trainer.synchronize()
        barrier()
        trainer.grad_transport, trainer.param_transport = args.grad_transport, args.param_transport
        bf16_leg = {"comm": comm16, "value": R * world * args.steps / elb, "unit": "rays/s", "ms_per_step": elb / args.steps * 1e3, "steps": args.steps,
                    "what": "same schedule and step as `value`, but the field-plane gradient is rounded to bf16 before the reduce-scatter and the parameter UPDATES "
                            "are all-gathered in bf16 (half the bytes on the xGMI links).  NOT the reference's fp32 DDP arithmetic: reported next to the headline, "
                            "never as it.  PSNR effect bounded on one GPU by emulation only (profiles/r03_psnr_30k_bf16_emulated_bf16_transports.json: "
                            "-0.03 +- 0.12 dB over 10 paired seeds; one rounding where a W-rank ring sum makes ~log2 W)"}
    steady_line = None
    if steady:
        trainer.synchronize()
        trainer.step, trainer._steps_since_update = max(args.start_step, 6000), 0
        for _ in range(max(args.warmup, 5)):
            one_step_steady()
        el2 = timed(one_step_steady, args.steps)
        steady_line = {"value": R * world * args.steps / el2, "unit": "rays/s", "ms_per_step": el2 / args.steps * 1e3, "steps": args.steps,
                       "schedule": f"optimiser steps {trainer.step - args.steps}..{trainer.step}: proposal networks updated every {cfg.proposal_update_every}th step",
                       "pixel_sampler": "DynamicBasedPixelSampler: 15 % of the rays from the IST maps (10 per image, without replacement), the rest uniform; "
                                        "maps + prefix sums precomputed outside the timed region",
                       "ist_nonzero_fraction": round(ist_fraction, 4)}
        trainer.synchronize()
        trainer.step, trainer._steps_since_update = args.start_step + args.warmup + args.steps, 0

    # the optimiser sweep runs on its own stream under the next step's first kernels (async_field_adam): its launch duration in the timed
    # region includes that sharing.  For context, time it ALONE as well (a few extra steps with the sweep back on the main stream).
    alone = None
    if trainer.async_field_adam and not trainer._sharded() and not args.no_alone:
        trainer.async_field_adam = False
        trainer.enable_kernel_timing(["adam_planes.field"])
        for _ in range(20):
            one_step()
        torch.cuda.synchronize()
        alone = trainer.kernel_times_ms()
        trainer.disable_kernel_timing()
        trainer.async_field_adam = True

    breakdown = None
    if args.breakdown:
        trainer.enable_kernel_timing(None)
        for _ in range(20):
            one_step()
        breakdown = trainer.kernel_times_ms()
        trainer.disable_kernel_timing()

    # trained state: the model REALLY trained up to --trained-until (default 5000; the reference's schedule: uniform pixels for 2000 steps,
    # then 15 % IST rays; cosine lr; proposal updates every 5th step from step 5000), untimed, then K timed steps.  BASELINE.json's metric is
    # quoted "past step 5000": with the density learnt the samples sit on the surfaces and a step costs less than on the untrained planes above.
    trained_line, breakdown_trained = None, None
    if steady and args.trained_until > 0:
        trainer.restart(init_params)  # fresh parameters, zero Adam moments, step 0: a clean run, not a continuation of the legs above
        while trainer.step < args.trained_until:
            batch["iter_steps"] = trainer.step
            one_step_steady()
        batch["iter_steps"] = trainer.step
        el3 = timed(one_step_steady, args.steps)
        if args.breakdown:
            trainer.enable_kernel_timing(None)
            for _ in range(20):
                one_step_steady()
            breakdown_trained = trainer.kernel_times_ms()
            trainer.disable_kernel_timing()
        trained_line = {"value": R * world * args.steps / el3, "unit": "rays/s", "ms_per_step": el3 / args.steps * 1e3, "steps": args.steps,
                        "what": f"the same steady-state schedule after {args.trained_until} real training steps FROM THE INITIAL PARAMETERS (trainer.restart: Adam "
                                "moments and step counters cleared) on the synthetic scene (untimed): density learnt, samples concentrated on the surfaces"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        S2 = cfg.num_nerf_samples_per_ray
        # algorithmic bytes / flops of ONE launch (SURVEY.md §8d conventions: 4 texels per bilinear tap, no cache credit)
        gather = R * S2 * len(cfg.multiscale_res) * 6 * 4 * cfg.feature_dim * 4  # every texel of every tap, once
        F = cfg.feature_dim * len(cfg.multiscale_res)
        fine_share = 1.0 - trainer._finest_offset() / max(trainer.field_planes.numel, 1)
        alg = {
            "adam_step": ("hbm", 32 * trainer.n_params, "adam_kernel: p,g,m,v read + p,m,v written + g cleared = 32 B/param"),
            "adam_planes.field": ("hbm", 32 * (trainer._field_seg[2] // world if trainer._sharded() else trainer.field_planes.numel) // sweep_launches,
                                  "plane_reg_kernel<32,true> (Adam + K-Planes regularisers fused, field planes): p,g,m,v read + p,m,v written + g cleared = 32 B/param"
                                  + (f"; {sweep_launches} launches per step (the finest scale's planes = {fine_share:.0%} of the floats, then the coarser scales', pipelined with "
                                     "pass B of the sorted scatter): bytes and duration are means over both, as rocprofv3's per-kernel average is" if sweep_launches > 1 else "")),
            "kplanes_scatter_sorted.field": ("hbm", 2 * gather, "pass B of the sorted scatter (scatter_halfwave_kernel<6,QUOT>): read-modify-write of every touched texel"),
            "kplanes_gather_bwd.field": ("hbm", 2 * gather, "kplanes_gather_bwd_kernel<32,6>: read-modify-write of every touched texel"),
            "kplanes_gradvec.field": ("hbm", gather + R * S2 * 30 * cfg.feature_dim * 4, "gradvec_kernel<32,6>: texel reads + per-plane gradient vectors written"),
            "kplanes_gather_fwd.field": ("hbm", gather, "kplanes_gather_fwd_kernel<32,6>: texel reads"),
            "kplanes_quotient_prepare": ("hbm", 3 * R * S2 * F * 4, "quotient_prepare_kernel: G = gfeat .* feat (two tensors read, one written)"),
            "kplanes_field_fwd": ("hbm", gather + R * S2 * (16 + 2 * F + 64),
                                  "field_fwd_kernel (gather + sigma_net + color_net fused): texel reads + density / rgb + the 16-bit feature tile and the 16 sigma_net outputs (kept for the backward) written"),
            # proposal planes (C = 8, one scale): two launches per updated step (256 and 128 samples per ray) -> mean bytes per launch; runs on
            # its own stream beside the field backward, so its launches are stretched by whatever shares the GPU with them
            "kplanes_gather_bwd.prop": ("hbm", 2 * R * (sum(cfg.num_proposal_samples_per_ray) // 2) * 6 * 4 * cfg.proposal_feature_dim * 4,
                                        "kplanes_gather_bwd_kernel<8,6>: read-modify-write of every touched proposal texel (mean of the two levels)"),
            "mlp_bwd.160x128x1": ("mfma", 3 * 2 * R * S2 * (F * cfg.sigma_net_hidden_dim + cfg.sigma_net_hidden_dim * 16), "mlp_bwd_kernel<160,128,1>: 3x forward flops (fp32 MFMA)"),
        }
        # kernels that run BESIDE the main chain on their own stream: their launch duration is stretched by whatever shares the GPU with
        # them (the proposal scatter: 0.35 ms per updated step alone, ~1.2 ms while gradvec / pass B run next to it), so elapsed time says
        # little about their cost; they are reported, not ranked
        SIDE = ("kplanes_gather_bwd.prop",)
        timed_k = {k: v for k, v in kt.items() if k in alg and k not in SIDE}
        side = {k: v for k, v in kt.items() if k in SIDE}
        per_step = lambda k: timed_k[k][0] * timed_k[k][1]
        DOMINANT = max(timed_k, key=per_step)  # the measured maximum; kernels within 10 % of it are listed under near_ties
        near_ties = [k for k in timed_k if k != DOMINANT and per_step(k) >= 0.9 * per_step(DOMINANT)]
        bound, alg_bytes, kdesc = alg[DOMINANT]
        dom_ms = timed_k[DOMINANT][0]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        peak = HBM_PEAK_GBS if bound == "hbm" else 157300.0  # fp32 MFMA peak, GFLOP/s
        # HBM traffic of that kernel from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately; profiles/r01_pmc_*):
        # counters cannot be read inside this process, so the committed per-launch figure for exactly this workload is quoted.
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_pmc_traffic.json")
        if os.path.exists(pmc) and R == 4096 and world == 1:
            traffic = json.load(open(pmc)).get(DOMINANT, {}).get("traffic_bytes_per_launch")
            if traffic is not None:
                traffic_source = (f"NOT measured in this run: quoted from profiles/{PROFILE_TAG}_pmc_traffic.json = 2 x FETCH_SIZE + WRITE_SIZE per launch of this "
                                  f"kernel, separate rocprofv3 --pmc passes of `bench.py --steps 6 --warmup 2 --images 38` (tools/collect_profiles.sh)")
        line = {
            "metric": "train rays/sec (K-Planes Broadcast-style, whole job)", "value": R * world * args.steps / elapsed, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.mlp_operands == "fp32" else args.mlp_operands,
            "dtype_note": "exact fp32 everywhere (the parity path)" if args.mlp_operands == "fp32" else
                          f"{args.mlp_operands} MFMA operands with f32 accumulation in every MLP (sigma_net, color_net, proposal nets); plane-gradient scatter in "
                          + ("its quotient form (f32)" if trainer.quotient_scatter else f"its product form with {cfg.gvec_dtype} gradient vectors between the two passes")
                          + "; planes, sampling, compositing, losses, gradient accumulation and the optimiser are f32",
            "data": "synthetic",
            "comm": comm_info,
            "config": {"workload": "K-Planes default multiscale-res 1-16 on synthetic Broadcast-style (k-planes preset: 4096 rays/GPU/step, "
                                   "samples 256/128/64, 5 scales x 6 planes C=32, 156.0 M params), full train step incl. Adam",
                       "schedule": "early training: proposal networks updated every 2nd step (every step for the first 10)" if args.start_step < 10 else
                                   f"from optimiser step {args.start_step} (proposal networks updated every {cfg.proposal_update_every}th step past step {cfg.proposal_warmup})",
                       "rays_per_gpu": R, "images": int(M), "image_hw": [int(H), int(W)], "params": int(trainer.n_params),
                       "parallelism": "single GPU" if world == 1 else (
                           f"ray-sharded x{world}; field-plane gradients: RCCL reduce-scatter ({trainer.grad_transport}) -> Adam on a 1/{world} shard -> all-gather of "
                           f"the {'parameter updates (bf16)' if trainer.param_transport == 'bf16' else 'new planes (fp32)'} (= one all-reduce's worth of "
                           "traffic), small segments: fp32 all-reduce" if trainer._sharded() else
                           f"ray-sharded x{world}, one RCCL all-reduce of the flat gradient buffer per step")},
            "roofline": {"bound": bound, "kernel": kdesc, "achieved": achieved, "peak": peak, "unit": "GB/s" if bound == "hbm" else "GFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_per_launch": alg_bytes,
                         "avg_launch_ms": dom_ms, "near_ties": near_ties,
                         **({"launch_breakdown": sweep_parts} if sweep_parts and DOMINANT == "adam_planes.field" else {}),
                         "launches_timed": timed_k[DOMINANT][1],
                         **({"alone": {"avg_launch_ms": round(alone[DOMINANT][0], 4),
                                       "frac": round(alg[DOMINANT][1] * (sweep_launches if DOMINANT == "adam_planes.field" else 1) / (alone[DOMINANT][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                       "note": "same kernel issued on the main stream as ONE launch over all field planes (20 extra steps after the timed region): nothing of the NEXT step runs beside it; "
                                               "in the timed region it shares the GPU with pass B of the coarser scales and the next step's first kernels"}}
                            if alone is not None and DOMINANT in alone else {}),
                         "other_kernels_ms": {k: round(v[0], 4) for k, v in sorted(timed_k.items(), key=lambda kv: -kv[1][0]) if k != DOMINANT},
                         "side_stream_kernels": {k: {"avg_launch_ms": round(v[0], 4), "launches_timed": v[1],
                                                     "frac": round(alg[k][1] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                                     "note": "runs beside the main chain on its own stream; duration stretched by co-running kernels "
                                                             "(alone: bench.py --breakdown --no-overlap)"} for k, v in side.items()},
                         "other_kernels_note": "fractions below use SURVEY 8d algorithmic bytes (4 texels per bilinear tap, no cache credit): a value "
                                               "above 1 means caches / run-length combining removed traffic, not that a peak was exceeded",
                         "other_kernels_frac": {k: round(alg[k][1] / (v[0] * 1e-3) / 1e9 / (HBM_PEAK_GBS if alg[k][0] == "hbm" else 157300.0), 3)
                                                for k, v in timed_k.items() if k != DOMINANT}},
        }
        if comm_timed is not None:
            line["comm"].update(comm_timed)
        if bf16_leg is not None:
            line["bf16_transports"] = bf16_leg
        if steady_line is not None:
            line["steady_state"] = steady_line
        if trained_line is not None:
            line["trained_state"] = trained_line
        # second half of BASELINE.json's metric (PSNR@30k): not re-measured here (a 30 k-step run takes minutes) -- the committed results of
        # tools/train_psnr.py on this workload are read from their files and quoted with their source
        psnr = {}
        # (r06) bf16 = the POOLED runs of the round-5 final code (tools/merge_psnr_runs.py: 16 seeds on the default scene, 12 on the textured one; the K-Planes
        # kernels have not changed since), not a 3-seed mid-round file; the textured scene's arms are quoted beside the default scene's
        for op in ("bf16", "fp32", "standin", "bf16_textured", "standin_textured"):
            # this round's file if it exists, else the last round's that does (the source is named in the line either way)
            f = next((c for c in (os.path.join(ROOT, "profiles", f"{tag}_psnr_30k_{op}.json") for tag in (PROFILE_TAG, "r05", "r04", "r03")) if os.path.exists(c)), None)
            if f is None:
                continue
            try:
                d = json.load(open(f))
                what = ("oracle/torch_standin.StandinTrainer = the reference's algorithm in stock PyTorch, fp32, same scene / sampler / schedule / evaluation"
                        if op.startswith("standin") else "tools/train_psnr.py: same preset, same synthetic scene")
                early = [r["stopped_early_at_step"] for r in d["runs"] if r.get("stopped_early_at_step")]
                psnr[op] = {"source": f"profiles/{os.path.basename(f)} ({what}, {len(d['runs'])} seed(s), eval frames per camera: {d.get('eval_frames', 'all')}"
                                      + (f"; NOT 30 000 steps: the runs were stopped at steps {early} by the one-hour limit of a GPU call and evaluated there" if early else "") + ")",
                            **{k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in d["summary"].items()}}
            except Exception as e:  # a malformed file must not take the bench line down
                psnr[op] = {"source": os.path.basename(f), "error": str(e)}
        if psnr:
            line["psnr_30k"] = psnr
        if world == 1 and not args.no_standin:
            trainer.synchronize()
            line["reference_standin"] = reference_standin(dev, R, line["value"])
        if world == 1 and not (args.no_config3 and args.no_config4):
            # the other single-GPU configurations of BASELINE.json, each a full train step of its own preset, timed like the headline (never `value`)
            trainer.synchronize()
            del trainer, init_params
            torch.cuda.empty_cache()
            for key, leg, skip in (("config3", config3_leg, args.no_config3), ("config4", config4_leg, args.no_config4)):
                if skip:
                    continue
                try:
                    line[key] = leg(dev, args)
                except Exception as e:  # a leg must not take the headline down; the failure is in the line
                    line[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(steps=args.cpu_steps)
        for title, bd in (("per-step kernel time by group", breakdown), ("trained state, per-step kernel time by group", breakdown_trained)):
            if not bd:
                continue
            tot = sum(v[0] * v[1] for v in bd.values()) / 20
            print(f"{title} (sum {tot:.3f} ms):", file=sys.stderr)
            for k, (ms, n) in sorted(bd.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                print(f"  {k:32s} {ms:8.3f} ms x {n / 20:4.1f}/step = {ms * n / 20:8.3f} ms", file=sys.stderr)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
