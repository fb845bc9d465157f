#!/bin/bash
# Run on the GPU box from the repo root: MFMA-pipe utilisation, LDS waits and occupancy of the MFMA kernels of the DEFAULT path (mlp_lp_fwd / bwd /
# bwd_tr, mlp_rows_bwd_kernel, sigma_bwd_kernel, field_fwd_kernel, density_fwd_kernel) -- separate --pmc passes, kernel-trace only.
# usage: bash tools/collect_mfma_pmc.sh <tag> [bench flags]     -> gpurun_out/<tag>_mfma_pmc.csv
set -u
set -o pipefail
TAG=${1:-r04}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n_ok=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU"; do
  t=$(echo $SET | tr ' ' '_' | cut -c1-40)
  if rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${TAG}_pmcm_$t -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-standin --no-config3 --no-config4 --no-alone --images 38 --no-steady-state "$@" > $OUT/${TAG}_pmcm_$t.log 2>&1 \
     && find $OUT/${TAG}_pmcm_$t -name '*counter_collection.csv' | grep -q .; then n_ok=$((n_ok + 1)); else echo "collect_mfma_pmc.sh: pass '$SET' failed (see $OUT/${TAG}_pmcm_$t.log)" >&2; fi
done
cd $ROOT
if [ $n_ok -lt 3 ]; then echo "collect_mfma_pmc.sh: fewer than three counter passes succeeded" >&2; exit 1; fi
TAG=$TAG ARGS="$*" python - <<'PY'
import csv, glob, collections, os
out, tag = "gpurun_out", os.environ["TAG"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/{tag}_pmcm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(p in k for p in ("mlp_lp_", "mlp_rows_bwd_kernel", "sigma_bwd_kernel", "field_fwd_kernel", "density_fwd_kernel")):
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
        "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VALU"]
with open(f"{out}/{tag}_mfma_pmc.csv", "w") as g:
    g.write(f"# rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-standin --no-config3 --no-config4 --no-alone --images 38 --no-steady-state {os.environ['ARGS']}   (one counter set per pass; mean per launch)\n")
    g.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the chip's matrix pipes busy while the kernel runs\n")
    g.write("# (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  wait_frac / lds_wait_frac / issue_frac = SQ_WAIT_ANY / SQ_WAIT_INST_LDS / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES\n")
    g.write("# (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES); waves_per_simd = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES x (SQ_BUSY_CYCLES counts per SE: a relative figure)\n")
    g.write("kernel," + ",".join(cols) + ",mfma_busy_frac,wait_frac,lds_wait_frac,issue_frac\n")
    for k, d in sorted(acc.items()):
        m = {c: (sum(d[c]) / len(d[c]) if d.get(c) else float("nan")) for c in cols}
        gui, wc = m["GRBM_GUI_ACTIVE"], m["SQ_WAVE_CYCLES"]
        frac = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * gui / 8) if gui == gui and gui else float("nan")
        per_wc = lambda c: m[c] / wc if wc == wc and wc else float("nan")  # a pass that reported no wave cycles gives nan, not a traceback
        g.write(f'"{k}",' + ",".join(f"{m[c]:.0f}" for c in cols) + f",{frac:.3f},{per_wc('SQ_WAIT_ANY'):.3f},{per_wc('SQ_WAIT_INST_LDS'):.3f},{per_wc('SQ_ACTIVE_INST_ANY'):.3f}\n")
print(open(f"{out}/{tag}_mfma_pmc.csv").read())
PY
rm -rf $OUT/${TAG}_pmcm_*
