"""How far can any other arithmetic follow G13b (the reference's own 50-step NeRFPlayer run) value by value?  Asked of the reference itself: its run
repeated (a) as committed -- must reproduce the fixture bit for bit, (b) with another ATen thread count (another summation order in reductions / GEMMs),
(c) from a start perturbed by one fp32 rounding (every parameter x (1 +- eps)).  Per step: the largest relative deviation of the loss terms, of PSNR and
the largest absolute deviation of the mean rendered probabilities from the committed run.  Build container only (imports the reference).  Dev tool.

    python tools/g13b_reference_spread.py --out profiles/r05_g13b_reference_spread.json
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["rgb_loss", "interlevel_loss", "distortion_loss", "temporal_tv_loss", "prob_loss"]


def run(tag, env):
    out = f"/tmp/g13b_{tag}.npz"
    e = dict(os.environ, SNERF_G13B_OUT=out, **env)
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden_nerfplayer_dynamics.py")], env=e, check=True, stdout=subprocess.DEVNULL)
    return np.load(out)


def deviation(ref, got):
    """Per step: {term: relative deviation (floor 1 % of the term's largest value in the run)}, probabilities: absolute."""
    steps = int(ref["steps"])
    rows = []
    for s in range(steps):
        row = {}
        for k in KEYS:
            c = ref["loss_" + k]
            floor = max(1e-2 * float(np.abs(c).max()), 1e-7)
            row[k] = abs(float(got["loss_" + k][s]) - float(c[s])) / max(abs(float(c[s])), floor)
        row["probs_abs"] = float(np.abs(got["probs_mean"][s] - ref["probs_mean"][s]).max())
        row["psnr_abs_db"] = abs(float(got["psnr"][s]) - float(ref["psnr"][s]))
        rows.append(row)
    return rows


def summary(rows):
    worst = [max(r[k] for k in KEYS) for r in rows]
    first = lambda tol: next((i for i, w in enumerate(worst) if w > tol), None)
    return {"first_step_with_a_loss_term_off_by_more_than": {"1e-3": first(1e-3), "1e-2": first(1e-2), "1e-1": first(1e-1)},
            "worst_loss_term_relative_deviation_by_step": [round(w, 6) for w in worst],
            "probs_abs_by_step": [round(r["probs_abs"], 5) for r in rows],
            "end_static_probability_deviation": round(rows[-1]["probs_abs"], 4),
            "worst_window_mean_probs_abs(steps 5..49)": round(max(float(np.mean([r["probs_abs"] for r in rows[lo:hi]])) for lo, hi in ((5, 14), (14, 23), (23, 32), (32, 41), (41, 50))), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    ref = np.load(os.path.join(ROOT, "tests", "golden", "g13b_nerfplayer_dynamics.npz"))
    res = {"what": __doc__.split("\n\n")[0]}
    same = run("same", {})
    res["rerun_bit_identical_to_fixture"] = bool(all(np.array_equal(ref[k], same[k]) for k in ref.files))
    for tag, env in (("threads_1", {"SNERF_G13B_THREADS": "1"}), ("threads_3", {"SNERF_G13B_THREADS": "3"}),
                     ("perturbed_1e-7", {"SNERF_G13B_PERTURB": "1e-7"}), ("perturbed_1e-7_b", {"SNERF_G13B_PERTURB": "-1e-7"}),
                     ("perturbed_1e-6", {"SNERF_G13B_PERTURB": "1e-6"})):
        res[tag] = summary(deviation(ref, run(tag, env)))
        print(tag, res[tag]["first_step_with_a_loss_term_off_by_more_than"], "end static dev", res[tag]["end_static_probability_deviation"])
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
