"""CPU: dev tools whose output is quoted by bench.py -- tools/merge_psnr_runs.py (per-seed PSNR files -> the one file the bench line reads)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_file(path, seed, novel, cam, train, ssim, steps=30000, scene="default"):
    ev = {"step": steps, "camera_20": {"psnr_mean": cam, "ssim_mean": 0.9}, "novel": {"psnr_mean": novel, "ssim_mean": ssim}, "train": {"psnr_mean": train, "ssim_mean": 0.9}}
    one = lambda v: {"mean": v, "min": v, "max": v, "std": 0.0, "n": 1}
    doc = {"config": "c", "steps": steps, "scene": scene, "eval_frames": 8, "trainer": "t", "runs": [{"seed": seed, "evals": [ev]}],
           "summary": {"camera_20": one(cam), "novel": one(novel), "train": one(train), "novel_ssim": one(ssim)}}
    json.dump(doc, open(path, "w"))


def test_merge_psnr_runs_recomputes_the_summary(tmp_path):
    a, b, c, out = (str(tmp_path / n) for n in ("a.json", "b.json", "c.json", "out.json"))
    _run_file(a, 3, 41.0, 39.0, 44.0, 0.99)
    _run_file(b, 4, 42.0, 37.0, 42.0, 0.97)
    _run_file(c, 5, 43.0, 38.0, 43.0, 0.98)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "merge_psnr_runs.py"), out, c, a, b], check=True, capture_output=True)
    d = json.load(open(out))
    assert [r["seed"] for r in d["runs"]] == [3, 4, 5] and d["merged_from"] == ["a.json", "b.json", "c.json"]
    s = d["summary"]["novel"]
    assert s["n"] == 3 and s["mean"] == pytest.approx(42.0) and s["min"] == 41.0 and s["max"] == 43.0 and s["std"] == pytest.approx(1.0)  # sample standard deviation
    assert d["summary"]["novel_ssim"]["mean"] == pytest.approx(0.98) and set(d["summary"]) == {"camera_20", "novel", "train", "novel_ssim"}


def test_merge_psnr_runs_refuses_mixed_or_duplicate_runs(tmp_path):
    a, b, c, out = (str(tmp_path / n) for n in ("a.json", "b.json", "c.json", "out.json"))
    _run_file(a, 3, 41.0, 39.0, 44.0, 0.99)
    _run_file(b, 3, 42.0, 37.0, 42.0, 0.97)                      # the same seed twice
    _run_file(c, 4, 42.0, 37.0, 42.0, 0.97, steps=26000)         # another run length
    tool = os.path.join(ROOT, "tools", "merge_psnr_runs.py")
    assert subprocess.run([sys.executable, tool, out, a, b], capture_output=True).returncode != 0
    assert subprocess.run([sys.executable, tool, out, a, c], capture_output=True).returncode != 0
    assert not os.path.exists(out)
