"""Merge per-seed result files of tools/train_psnr.py (one 30 000-step run per GPU call: tools/run_standin_30k.sh) into one file of the same format, with
the summary (mean / min / max / std / n per evaluation set) recomputed over all runs.  bench.py quotes profiles/<round>_psnr_30k_<arm>.json.  Dev tool.

    python tools/merge_psnr_runs.py profiles/r05_psnr_30k_standin.json profiles/r05_psnr_30k_standin_default_seed*.json
"""
import json
import sys


def stats(xs):
    """tools/train_psnr.py's statistics (sample standard deviation)."""
    n = len(xs)
    mean = sum(xs) / n
    return {"mean": mean, "min": min(xs), "max": max(xs), "std": (sum((x - mean) ** 2 for x in xs) / max(n - 1, 1)) ** 0.5, "n": n}


def summarise(runs):
    """Over the runs' LAST evaluation, per evaluation set, as tools/train_psnr.py writes it."""
    final = [r["evals"][-1] for r in runs]
    out = {name: stats([f[name]["psnr_mean"] for f in final]) for name in final[0] if isinstance(final[0][name], dict)}
    out["novel_ssim"] = stats([f["novel"]["ssim_mean"] for f in final])
    return out


def main():
    dst, srcs = sys.argv[1], sorted(sys.argv[2:])
    docs = [json.load(open(s)) for s in srcs]
    head = {k: v for k, v in docs[0].items() if k not in ("runs", "summary")}
    for d, s in zip(docs[1:], srcs[1:]):
        for k in ("steps", "scene", "eval_frames", "trainer"):
            if d.get(k) != head.get(k):
                raise SystemExit(f"{s}: {k} = {d.get(k)!r} differs from {srcs[0]}: {head.get(k)!r}")
    runs = sorted((r for d in docs for r in d["runs"]), key=lambda r: r["seed"])
    if len({r["seed"] for r in runs}) != len(runs):
        raise SystemExit("the same seed appears in two files")
    head["merged_from"] = [s.split("/")[-1] for s in srcs]
    ref = docs[0]["summary"]
    new = summarise(runs)
    # the recomputed summary of the FIRST file alone must reproduce the one train_psnr.py wrote (guards against the two drifting apart)
    one = summarise(docs[0]["runs"])
    for k, v in ref.items():
        if abs(one[k]["mean"] - v["mean"]) > 1e-9 or abs(one[k]["std"] - v["std"]) > 1e-9:
            raise SystemExit(f"summary key {k}: recomputed {one[k]} != recorded {v}")
    head["runs"], head["summary"] = runs, {k: new[k] for k in ref if k in new}
    json.dump(head, open(dst, "w"), indent=1)
    print(dst, {k: (round(v["mean"], 3), round(v["std"], 3), v["n"]) for k, v in head["summary"].items()})


if __name__ == "__main__":
    main()
