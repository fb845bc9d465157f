"""CPU, world_size 2, gloo: the N>1 path of the hot path.

(1) the single collective: SUM all-reduce of a flat gradient buffer + 1/world scale == DDP's mean;
(2) semantics: two ranks with R rays each, gradients averaged == one rank with the 2R-ray batch (every data loss is
    a mean over rays; parameter-only regularisers are identical on every rank) -- gradients from the CPU oracle;
(3) per-rank seeding (seed + rank) gives different rays on different ranks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _small_params():
    from oracle import kplanes_oracle as KO

    return KO.make_kplanes_params(base_res=(8, 8, 8, 3), multiscale=(1, 2), feat_dim=32, prop_res=((12, 12, 12, 3), (16, 16, 16, 3)),
                                  prop_feat=8, sigma_hidden=128, color_hidden=64, seed=11)


def _batch(R, seed):
    gen = torch.Generator().manual_seed(seed)
    o = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.2
    d = torch.nn.functional.normalize(torch.rand(R, 3, generator=gen) * 2 - 1, dim=-1)
    rays = {"origins": o, "directions": d, "times": torch.rand(R, 1, generator=gen)}
    rng = {"t_rand": torch.rand(R, 33, generator=gen), "u": [torch.rand(R, 17, generator=gen), torch.rand(R, 9, generator=gen)],
           "bg": torch.rand(R, 3, generator=gen)}
    return rays, rng, torch.rand(R, 3, generator=gen)


def _flat_grad(P, rays, rng, target):
    from oracle import kplanes_oracle as KO

    leaves = KO.all_param_tensors(P)
    for x in leaves:
        x.requires_grad_(True)
        x.grad = None
    out = KO.kplanes_forward(P, rays, rng, (32, 16), 8, anneal=0.5)
    sum(KO.kplanes_loss_dict(P, out, target).values()).backward()
    return torch.cat([(x.grad if x.grad is not None else torch.zeros_like(x)).reshape(-1) for x in leaves])


def _worker(rank, world, port, R, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from soccernerfs_amd import dist as sdist

    r, w, pg = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rays, rng, target = _batch(R, sdist.rank_seed(100, rank))
    g = _flat_grad(_small_params(), rays, rng, target)
    local = g.clone()
    scale = sdist.allreduce_flat_(g, pg)
    assert scale == 0.5
    t = sdist.max_over_ranks(float(rank + 1), "cpu", pg)
    if rank == 0:
        torch.save((local, g * scale, rays["origins"], t), os.path.join(outdir, "r0.pt"))
    else:
        torch.save((local, rays["origins"]), os.path.join(outdir, "r1.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_mean_equals_double_batch(tmp_path):
    R = 12
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, R, str(tmp_path))) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    local0, reduced, o0, tmax = torch.load(tmp_path / "r0.pt")
    local1, o1 = torch.load(tmp_path / "r1.pt")
    assert tmax == 2.0
    assert not torch.equal(o0, o1)  # seed + rank => different rays
    torch.testing.assert_close(reduced, (local0 + local1) / 2, rtol=1e-6, atol=1e-9)
    # one rank, both batches concatenated
    (ra, ga, ta), (rb, gb, tb) = _batch(R, 100), _batch(R, 101)
    cat = lambda a, b: torch.cat([a, b], 0)
    rays = {k: cat(ra[k], rb[k]) for k in ra}
    rng = {"t_rand": cat(ga["t_rand"], gb["t_rand"]), "u": [cat(ga["u"][i], gb["u"][i]) for i in range(2)], "bg": cat(ga["bg"], gb["bg"])}
    full = _flat_grad(_small_params(), rays, rng, cat(ta, tb))
    torch.testing.assert_close(reduced, full, rtol=2e-4, atol=1e-7)


def _shard_worker(rank, world, port, outdir):
    """Reduce-scatter -> 'optimiser' on the shard -> all-gather must equal all-reduce -> optimiser on everything (dist.py)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from soccernerfs_amd import dist as sdist

    _, _, pg = sdist.init_from_env(backend="gloo")
    n = 4 * world * 37
    g = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank
    p = torch.linspace(-1, 1, n)
    shard = torch.empty(n // world)
    sdist.reduce_scatter_sum(shard, g, pg, async_op=True).wait()
    lo = rank * (n // world)
    new_shard = p[lo:lo + n // world] - 0.1 * shard / world  # an SGD step stands in for Adam
    full = torch.empty(n)
    sdist.all_gather_shards(full, new_shard.contiguous(), pg, async_op=True).wait()
    ref_g = g.clone()
    sdist.all_reduce_sum_(ref_g, pg, async_op=True).wait()
    # half-width transports (opt-in): gradients rounded to bf16 before the reduce-scatter, parameter UPDATES gathered in bf16
    # and applied by every rank -- the owner of a shard included -- to its OLD parameters
    g16 = (torch.sin(torch.arange(n, dtype=torch.float32) * 0.37 + rank) * 3.0).to(torch.bfloat16)
    shard16 = torch.empty(n // world, dtype=torch.bfloat16)
    sdist.reduce_scatter_sum(shard16, g16, pg, async_op=True).wait()
    upd = (-0.1 * shard16.float() / world).to(torch.bfloat16)
    upd_full = torch.empty(n, dtype=torch.bfloat16)
    sdist.all_gather_shards(upd_full, upd, pg, async_op=True).wait()
    new16 = torch.add(p, upd_full)  # fp32 + bf16 -> fp32, the same expression on every rank
    g16_all = [torch.empty_like(g16) for _ in range(world)]
    dist.all_gather(g16_all, g16, group=pg)
    want_sum = sum(t.float() for t in g16_all)  # exact sum of the rounded gradients
    torch.save((full, p - 0.1 * ref_g / world, new16, shard16.float(), want_sum[lo:lo + n // world]), os.path.join(outdir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_reduce_scatter_shard_all_gather_equals_all_reduce(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(rk, 2, port, str(tmp_path))) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    (f0, r0, n0, s0, w0), (f1, r1, n1, s1, w1) = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    assert torch.equal(f0, f1) and torch.equal(r0, r1)
    torch.testing.assert_close(f0, r0, rtol=0, atol=0)
    assert n0.dtype == torch.float32 and torch.equal(n0, n1)  # replicas stay bit-identical with the bf16 update transport
    for s_, w_ in ((s0, w0), (s1, w1)):  # bf16 reduce-scatter: the sum of two bf16 values, rounded once
        torch.testing.assert_close(s_, w_, rtol=2 ** -8, atol=1e-6)


def _chunk_worker(rank, world, port, outdir):
    """The chunked exchange of trainer._plan_exchange / _sharded_optimizer_step on plain tensors: the segment is cut in two chunks (tail
    first), each reduce-scattered, stepped on this rank's shard of THAT chunk and all-gathered -- must equal all-reduce + full step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from soccernerfs_amd import dist as sdist

    _, _, pg = sdist.init_from_env(backend="gloo")
    q = 4 * world
    n, cut = q * 50, q * 13
    g = torch.cos(torch.arange(n, dtype=torch.float32) * 0.11 + rank) * (rank + 2)
    p = torch.linspace(-1, 1, n)
    new = torch.zeros(n)
    works = []
    chunks = [(cut, n), (0, cut)]  # finest (tail) first
    shards = [torch.empty((hi - lo) // world) for lo, hi in chunks]
    for (lo, hi), sh in zip(chunks, shards):  # both reduce-scatters in flight before either is consumed
        works.append(sdist.reduce_scatter_sum(sh, g[lo:hi], pg, async_op=True))
    owned = torch.zeros(n, dtype=torch.bool)
    for (lo, hi), sh, w in zip(chunks, shards, works):
        w.wait()
        k = (hi - lo) // world
        a = lo + rank * k
        owned[a:a + k] = True
        sdist.all_gather_shards(new[lo:hi], (p[a:a + k] - 0.1 * sh / world).contiguous(), pg, async_op=True).wait()
    ref = g.clone()
    sdist.all_reduce_sum_(ref, pg)
    torch.save((new, p - 0.1 * ref / world, owned), os.path.join(outdir, f"c{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_chunked_exchange_equals_all_reduce(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(rk, 2, port, str(tmp_path))) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    (n0, r0, o0), (n1, r1, o1) = torch.load(tmp_path / "c0.pt"), torch.load(tmp_path / "c1.pt")
    assert torch.equal(n0, n1) and torch.equal(n0, r0)
    assert bool((o0 ^ o1).all())  # the two ranks' shards tile the segment: every float has exactly one owner


def test_single_process_is_identity():
    from soccernerfs_amd import dist as sdist

    g = torch.arange(8.0)
    assert sdist.allreduce_flat_(g, None) == 1.0 and torch.equal(g, torch.arange(8.0))
    assert sdist.rank_seed(7, 3) == 10
