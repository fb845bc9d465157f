"""GPU tests of the multi-GPU optimiser path (dist.py): reduce-scatter -> Adam on a 1/world shard -> all-gather.

The GPU box has ONE device, so the two-rank run uses two processes that share cuda:0 and exchange through gloo (dist.py stages
gloo collectives through host memory); the RCCL calls themselves are exercised with a world-size-1 nccl group."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_cfg():
    from soccernerfs_amd.trainer import KPlanesTrainConfig

    cfg = KPlanesTrainConfig(spacetime_resolution=(16, 16, 16, 5), multiscale_res=(1, 2), proposal_resolutions=((24, 24, 24, 5), (32, 32, 32, 5)), mlp_operands="fp32")
    cfg.num_proposal_samples_per_ray, cfg.num_nerf_samples_per_ray = (64, 32), 16
    cfg.warm_up_end = 2
    return cfg


def _batch(R, rank, step):
    gen = torch.Generator().manual_seed(1000 * rank + step)
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    dv = lambda z: z.to(DEV).contiguous()
    rays = {"origins": dv(o), "directions": dv(d), "times": dv(torch.rand(R, 1, generator=gen))}
    target = dv(torch.rand(R, 3, generator=gen))
    rng = {"t_rand": dv(torch.rand(R, 65, generator=gen)), "u": [dv(torch.rand(R, 33, generator=gen)), dv(torch.rand(R, 17, generator=gen))],
           "bg": dv(torch.rand(R, 3, generator=gen))}
    return rays, target, rng


def test_adam_planes_range_shards_equal_whole_sweep():
    """Three ranges that tile the segment, launched one after the other, give bit-identical p, m, v to the single sweep, and the
    regulariser partial sums add up to the whole."""
    from soccernerfs_amd import ops
    from soccernerfs_amd.plane_set import PlaneSet

    gen = torch.Generator(device=DEV).manual_seed(3)
    ps = PlaneSet(32, [[20, 12, 9, 7], [40, 24, 18, 7]], concat=True, device=DEV)
    n = ps.numel
    rnd = lambda: torch.rand(n, device=DEV, generator=gen)
    p, g, m, v = rnd() + 0.5, rnd() - 0.5, (rnd() - 0.5) * 0.1, rnd() * 0.01
    coefs = (0.3, 0.2, 0.1)

    def run(ranges):
        po, mm, vv = torch.zeros_like(p), m.clone(), v.clone()
        losses = torch.zeros(ops.REG_SLOTS, 16, device=DEV)
        for r in ranges:
            ops.adam_planes_step(ps, p, po, g.clone(), mm, vv, coefs, losses, 3, 1e-2, grad_scale=0.5, zero_grad=False, shard_range=r)
        return po, mm, vv, losses[:, :3].double().sum(0)

    whole = run([None])
    cuts = [0, (n // 3) // 4 * 4 + 4, (2 * n // 3) // 4 * 4, (n + 3) // 4 * 4]
    parts = run([(cuts[i], cuts[i + 1]) for i in range(3)])
    for a, b in zip(whole[:3], parts[:3]):
        assert torch.equal(a, b)
    torch.testing.assert_close(parts[3], whole[3], rtol=1e-6, atol=0)
    # a range touches nothing outside itself
    po, mm, vv, _ = run([(cuts[1], cuts[2])])
    assert float(po[:cuts[1]].abs().max()) == 0.0 and float(po[cuts[2]:].abs().max()) == 0.0
    assert torch.equal(mm[:cuts[1]], m[:cuts[1]]) and torch.equal(vv[cuts[2]:], v[cuts[2]:])


def _reference_two_rank_steps(n_steps, R):
    """Single process: per step, the gradients of rank 0's and rank 1's batches are summed and Adam applies their mean."""
    from soccernerfs_amd.trainer import KPlanesTrainer, anneal_value, update_schedule

    cfg = _small_cfg()
    tr = KPlanesTrainer(cfg, R, DEV)
    for k in range(n_steps):
        anneal = anneal_value(tr.step, cfg.proposal_weights_anneal_max_num_iters, cfg.proposal_weights_anneal_slope)
        sstep = max(tr.step - 1, 0)
        updated = tr._steps_since_update > update_schedule(sstep, cfg.proposal_warmup, cfg.proposal_update_every) or sstep < 10
        total = torch.zeros_like(tr.grads)
        for rank in range(2):
            rays, target, rng = _batch(R, rank, k)
            tr.forward(rays, rng, anneal, training=True)
            tr.backward(target, rng, proposal_grads=updated, include_reg=False)
            torch.cuda.synchronize()
            total += tr.grads
            tr.grads.zero_()
        tr.grads.copy_(total)
        tr._grad_scale = 0.5
        tr.optimizer_step(fused_reg=True)
        if updated:
            tr._steps_since_update = 0
        tr._steps_since_update += 1
    torch.cuda.synchronize()
    return tr


def _owner_mask(tr, rank, world=2):
    """Which floats of the field-plane segment rank `rank` owns under the chunked exchange plan (trainer._plan_exchange)."""
    from soccernerfs_amd.trainer import KPlanesTrainer

    probe = KPlanesTrainer.__new__(KPlanesTrainer)
    probe.cfg, probe.world, probe.dev = tr.cfg, world, torch.device("cpu")
    _, n, _ = tr._field_seg
    q = 4 * world
    probe._field_seg = (0, n, (n + q - 1) // q * q)
    probe._desc_field = tr._desc_field
    probe._plan_exchange()
    mask = torch.zeros(probe._field_seg[2], dtype=torch.bool)
    for ch in probe._exchange:
        mask[ch["lo"] + rank * ch["shard"]:ch["lo"] + (rank + 1) * ch["shard"]] = True
    return mask


def _worker_main():
    """python tests/test_gpu_sharded.py <rank> <world> <port> <steps> <R> <outdir> <shard 0|1|2>   (2 = sharded with bf16 gradient transport)"""
    rank, world, port, n_steps, R = (int(x) for x in sys.argv[1:6])
    outdir, shard = sys.argv[6], int(sys.argv[7])
    transport = "bf16" if shard == 2 else "fp32"
    shard = bool(shard)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from soccernerfs_amd.trainer import KPlanesTrainer

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    tr = KPlanesTrainer(_small_cfg(), R, DEV, process_group=dist.group.WORLD)
    tr.shard_optimizer = shard
    tr.grad_transport = transport
    tr.param_transport = transport
    for k in range(n_steps):
        rays, target, rng = _batch(R, rank, k)
        tr.train_step(rays, target, rng)
    tr.synchronize()
    reg = tr.loss_dict()
    if transport == "bf16":
        torch.save({"segments": {name: tr.views[name].cpu() for name in tr.views}, "step": tr.step}, os.path.join(outdir, f"rank{rank}_bf16.pt"))
        dist.destroy_process_group()
        return
    snap = {"segments": {name: tr.views[name].cpu() for name in tr.views}, "m": {name: tr.mviews[name].cpu() for name in tr.mviews},
            "gmax": float(tr.grads.abs().max()), "space_tv": float(reg["space_tv_loss"]), "step": tr.step}
    ck = tr.save_checkpoint(os.path.join(outdir, f"ckpt{int(shard)}"))  # collective when the optimiser is sharded (gathers the moment shards)
    assert (ck is not None) == (rank == 0)
    torch.save(snap, os.path.join(outdir, f"rank{rank}_{int(shard)}.pt"))
    dist.destroy_process_group()


def KPlanesTrainerForResume(tmp_path, shard):
    from soccernerfs_amd.trainer import KPlanesTrainer

    tr = KPlanesTrainer(_small_cfg(), 48, DEV)
    tr.load_checkpoint(os.path.join(tmp_path, f"ckpt{int(shard)}"))
    tr.synchronize()
    return tr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("shard", [True, False])
def test_two_ranks_on_one_gpu_match_mean_gradient_reference(tmp_path, shard):
    n_steps, R = 3, 48
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", str(port), str(n_steps), str(R), str(tmp_path), str(int(shard))],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [torch.load(os.path.join(tmp_path, f"rank{r}_{int(shard)}.pt")) for r in range(2)]
    ref = _reference_two_rank_steps(n_steps, R)
    assert res[0]["step"] == res[1]["step"] == n_steps
    assert res[0]["gmax"] == 0.0 and res[1]["gmax"] == 0.0  # gradient buffers cleared for the next step
    for name in ref.views:
        a, b = res[0]["segments"][name], res[1]["segments"][name]
        assert torch.equal(a, b), f"ranks disagree on {name}"  # replicas stay bit-identical
        want = ref.views[name].cpu()
        # Adam's first steps move a parameter by ~lr * sign(g): where g is rounding noise around 0 the atomic order can flip it
        bad = ((a - want).abs() > 2e-4).float().mean()
        assert float(bad) < 2e-3, (name, float(bad))
        m_ref = ref.mviews[name].cpu()
        m_got = res[0]["m"][name]
        if shard and name == "field.planes":  # every rank keeps Adam state for its own half of each exchange chunk only
            own0 = _owner_mask(ref, rank=0)[:m_ref.numel()]
            assert float(res[0]["m"][name][~own0].abs().max()) == 0.0 and float(res[1]["m"][name][own0].abs().max()) == 0.0
            m_got = torch.where(own0, res[0]["m"][name], res[1]["m"][name])
        assert float(m_ref.abs().max()) > 0
        torch.testing.assert_close(m_got, m_ref, rtol=0, atol=2e-3 * float(m_ref.abs().max()))
    want_tv = float(ref.loss_dict()["space_tv_loss"])
    for r in range(2):
        assert abs(res[r]["space_tv"] - want_tv) <= 2e-3 * abs(want_tv) + 1e-9
    # the checkpoint rank 0 wrote holds the WHOLE Adam state (the shards are gathered first) under the reference's names
    from soccernerfs_amd import checkpoint as CK

    files = os.listdir(os.path.join(tmp_path, f"ckpt{int(shard)}"))
    assert files == [f"step-{n_steps - 1:09d}.ckpt"]
    ck = torch.load(os.path.join(tmp_path, f"ckpt{int(shard)}", files[0]), map_location="cpu", weights_only=False)
    moments = CK.import_optimizer_states(ref._named_module(), ck["optimizers"])
    m_ck = moments["field.grids.planes"][0].reshape(-1)
    m_ref = ref.mviews["field.planes"].cpu()
    own0 = _owner_mask(ref, rank=0)[:m_ref.numel()]
    assert float(m_ck[own0].abs().max()) > 0 and float(m_ck[~own0].abs().max()) > 0
    torch.testing.assert_close(m_ck, m_ref, rtol=0, atol=2e-3 * float(m_ref.abs().max()))
    resumed = KPlanesTrainerForResume(tmp_path, shard)
    assert resumed.step == n_steps
    for name in ref.views:
        assert torch.equal(resumed.views[name].cpu(), res[0]["segments"][name]), name


def test_sharded_step_through_rccl_world_size_one():
    """The real RCCL entry points (reduce_scatter_tensor / all_gather_into_tensor / all_reduce, asynchronous) with a one-rank group:
    the sharded step must equal the plain fused step bit for bit (no atomics-order difference: same kernels, same order)."""
    import torch.distributed as dist
    from soccernerfs_amd.trainer import KPlanesTrainer

    R, n_steps = 48, 3
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        tr = KPlanesTrainer(_small_cfg(), R, DEV, process_group=dist.group.WORLD)
        assert tr.world == 1 and not tr._sharded()
        tr.world, tr.shard_optimizer = 1, True
        # force the sharded code path with a single shard per exchange chunk
        tr._sharded = lambda: True
        tr._plan_exchange()
        assert len(tr._exchange) == 2 and tr._exchange[0]["hi"] == tr._field_seg[2]  # finest scale first
        for k in range(n_steps):
            tr.train_step(*_batch(R, 0, k)[:2], _batch(R, 0, k)[2])
        tr.synchronize()
        got = {n: v.clone() for n, v in tr.views.items()}
        tv = float(tr.loss_dict()["space_tv_loss"])
        # the same through RCCL with the bf16 transports (bench.py's default for N > 1): bf16 reduce_scatter_tensor / all_gather_into_tensor
        t16 = KPlanesTrainer(_small_cfg(), R, DEV, process_group=dist.group.WORLD)
        t16.world, t16.shard_optimizer = 1, True
        t16._sharded = lambda: True
        t16._plan_exchange()
        t16.grad_transport = t16.param_transport = "bf16"
        for k in range(n_steps):
            t16.train_step(*_batch(R, 0, k)[:2], _batch(R, 0, k)[2])
        t16.synchronize()
        got16 = {n: v.clone() for n, v in t16.views.items()}
    finally:
        dist.destroy_process_group()
    ref = KPlanesTrainer(_small_cfg(), R, DEV)
    for k in range(n_steps):
        ref.train_step(*_batch(R, 0, k)[:2], _batch(R, 0, k)[2])
    ref.synchronize()
    for name in ref.views:
        bad = ((got[name] - ref.views[name]).abs() > 2e-4).float().mean()
        assert float(bad) < 2e-3, (name, float(bad))
    assert abs(tv - float(ref.loss_dict()["space_tv_loss"])) <= 2e-3 * abs(tv) + 1e-9
    # bf16 transports: Adam's normalised step barely feels a 2^-9 relative rounding of the gradient, except where a tiny gradient changes sign
    for name in ref.views:
        d = (got16[name] - ref.views[name]).abs()
        assert float(d.mean()) < 3e-4 and bool(torch.isfinite(got16[name]).all()), (name, float(d.mean()))


if __name__ == "__main__":
    _worker_main()


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_script_two_ranks_on_one_gpu(tmp_path, launcher):
    """bench.py for N = 2 -- as the driver launches it (torch.distributed.run, one process per rank) and as a bare `python bench.py --gpus 2`
    (the script starts its own ranks as child processes) -- with both ranks on device 0 and gloo collectives: the N > 1 branches of the
    script (sharded optimiser, MAX over ranks, JSON line with the link-byte / exposed-communication report) run end to end."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, SNERF_BENCH_ONE_DEVICE="1", SNERF_BENCH_BACKEND="gloo")
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--images", "38", "--trained-until", "6"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["steps"] == 3
    assert line["comm"]["ranks_counted_by_all_reduce"] == 2
    lb = line["comm"]["link_bytes_per_step_per_gpu"]
    npad = line["config"]["params"]  # the field planes are ~98 % of it: reduce-scatter + all-gather of fp32 = (W-1)/W x 4 B each way
    assert 0.9 * 0.5 * 4 * npad < lb["reduce_scatter.field"] <= 0.5 * 4 * npad and lb["all_gather.field"] == lb["reduce_scatter.field"]
    assert lb["total"] == sum(v for k, v in lb.items() if k != "total")
    assert line["comm"]["exposed_ms_per_step"] >= 0 and "comm_wait.reduce_scatter" in line["comm"]["exposed_by_wait"]
    assert line["bf16_transports"]["comm"]["link_bytes_per_step_per_gpu"]["reduce_scatter.field"] * 2 == lb["reduce_scatter.field"]
    # the headline is the reference's DDP arithmetic (fp32 on the links); the bf16 transports are a labelled second leg of the same process
    assert "reduce-scatter (fp32)" in line["config"]["parallelism"] and "new planes (fp32)" in line["config"]["parallelism"]
    assert line["bf16_transports"]["value"] > 0 and line["bf16_transports"]["steps"] == 3 and "NOT the reference" in line["bf16_transports"]["what"]
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    assert line["steady_state"]["value"] > 0 and line["trained_state"]["value"] > 0 and line["trained_state"]["steps"] == 3


def test_bf16_gradient_transport_two_ranks(tmp_path):
    """Opt-in bf16 transport of the field-plane gradient and of the parameter updates (bench.py --grad-transport bf16 --param-transport
    bf16): replicas stay bit-identical to each other and the
    parameters stay close to the fp32-transport reference (Adam's normalised step is insensitive to 2^-9 relative gradient rounding except
    where a gradient is rounding noise around zero)."""
    n_steps, R = 3, 48
    port = _free_port()
    env = dict(os.environ, PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), "2", str(port), str(n_steps), str(R), str(tmp_path), "2"],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = [torch.load(os.path.join(tmp_path, f"rank{r}_bf16.pt")) for r in range(2)]
    ref = _reference_two_rank_steps(n_steps, R)
    assert res[0]["step"] == res[1]["step"] == n_steps
    for name in ref.views:
        a, b = res[0]["segments"][name], res[1]["segments"][name]
        assert torch.equal(a, b), f"ranks disagree on {name}"
        want = ref.views[name].cpu()
        bad = ((a - want).abs() > 2e-4).float().mean()
        assert float(bad) < 2e-2, (name, float(bad))


def test_cabi_communicator_world_size_one():
    """snerf_comm_* / snerf_allreduce_grads (the gradient exchange behind the C ABI, include/snerf.h): a one-rank RCCL communicator created from
    a unique id; the in-place SUM all-reduce of one rank leaves the buffer unchanged and runs on the caller's stream."""
    import ctypes as C

    from soccernerfs_amd import _lib, ops

    L = _lib.lib()
    ident = (C.c_ubyte * 128)()
    _lib.check(L.snerf_comm_unique_id(ident), "comm_unique_id")
    assert any(ident)  # RCCL filled it
    comm = C.c_void_p()
    _lib.check(L.snerf_comm_create(1, 0, ident, C.byref(comm)), "comm_create")
    try:
        g = torch.randn(1 << 20, device=DEV)
        want = g.clone()
        _lib.check(L.snerf_allreduce_grads(comm, ops._ptr(g), C.c_int64(g.numel()), ops._stream()), "allreduce_grads")
        torch.cuda.synchronize()
        assert torch.equal(g, want)
        assert L.snerf_allreduce_grads(None, ops._ptr(g), C.c_int64(4), ops._stream()) < 0  # argument error, message set
        assert b"communicator" in L.snerf_last_error()
    finally:
        _lib.check(L.snerf_comm_destroy(comm), "comm_destroy")
