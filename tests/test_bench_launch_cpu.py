"""bench.py's own rank launcher (no GPU): `python bench.py --gpus N` with no launcher around it must start N ranks itself (the
reference spawns its own ranks: NSR/scripts/train.py:187-200, NCCL init :124-137) and must FAIL, not shrink, when it cannot.

The SNERF_BENCH_RANK_CHECK_ONLY hook stops every rank after the rendezvous + rank-count all-reduce (gloo), before anything touches a
GPU, so the launcher itself is covered here; the full two-rank bench line is covered on the GPU box by
tests/test_gpu_sharded.py::test_bench_script_self_launch_two_ranks_on_one_gpu."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(PYTHONPATH=ROOT, **env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)


def test_gpus_2_without_launcher_starts_two_ranks():
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], SNERF_BENCH_ONE_DEVICE="1", SNERF_BENCH_BACKEND="gloo", SNERF_BENCH_RANK_CHECK_ONLY="1")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_counted_by_all_reduce"] == 2
    assert "starting 2 ranks" in out.stderr and "torch.distributed.run" in out.stderr


def test_short_rank_count_is_an_error_not_a_smaller_job():
    # (a) a launcher that started fewer ranks than --gpus asks for
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", SNERF_BENCH_RANK_CHECK_ONLY="1")
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # (b) a rank that joined the rendezvous but is not counted by the all-reduce
    out = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], SNERF_BENCH_ONE_DEVICE="1", SNERF_BENCH_BACKEND="gloo", SNERF_BENCH_RANK_CHECK_ONLY="1",
               SNERF_BENCH_TEST_DROP_RANK="1")
    assert out.returncode != 0 and "counted by the all-reduce" in out.stderr, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_more_gpus_than_devices_refuses_before_spawning():
    # this container has no HIP device: the real (nccl) path must refuse `--gpus 8` loudly instead of printing n_gpus: 1
    out = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], HIP_VISIBLE_DEVICES="")
    assert out.returncode == 4 and "HIP device(s) visible" in out.stderr, out.stderr[-2000:]
    assert "starting 8 ranks" not in out.stderr


def test_launcher_without_gpus_flag_takes_the_launchers_rank_count():
    # `torchrun --nproc-per-node 2 bench.py` with no --gpus: the launcher's WORLD_SIZE is the request; no abort, n_gpus = 2
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(PYTHONPATH=ROOT, SNERF_BENCH_ONE_DEVICE="1", SNERF_BENCH_BACKEND="gloo", SNERF_BENCH_RANK_CHECK_ONLY="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          "29653", BENCH, "--steps", "1", "--warmup", "0"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # and with no launcher and no flag: one rank
    out = _run(["--steps", "1", "--warmup", "0"], SNERF_BENCH_RANK_CHECK_ONLY="1")
    assert out.returncode == 0 and json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
