#!/bin/bash
# Run on the GPU box from the repo root: MFMA-pipe utilisation of the MLP kernels (separate --pmc passes, kernel-trace only).
# usage: bash tools/collect_mfma_pmc.sh <tag> [bench flags]     -> gpurun_out/<tag>_mfma_pmc.csv
set -u
TAG=${1:-r01}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmcm_$C -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --images 38 "$@" > /dev/null 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
out, tag = "gpurun_out", "$TAG"
acc = collections.defaultdict(dict)
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"):
    f = glob.glob(f"{out}/{tag}_pmcm_{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    tmp = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c and "mlp" in r["Kernel_Name"]:
            tmp[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in tmp.items():
        acc[k][c] = sum(v) / len(v)
with open(f"{out}/{tag}_mfma_pmc.csv", "w") as g:
    g.write("# rocprofv3 --pmc <C> --kernel-trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --images 38 $*   (one counter per pass; mean per launch)\n")
    g.write("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the chip's matrix pipes busy while the kernel\n# runs (GRBM_GUI_ACTIVE is summed over the 8 XCDs; cross-check: sigma_net backward issues 16.4 M v_mfma_f32_16x16x4_f32 x 32 cycles = 524 M)\n")
    g.write("kernel,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,GRBM_GUI_ACTIVE,mfma_busy_frac\n")
    for k, v in sorted(acc.items()):
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        frac = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * gui / 8) if gui else float("nan")
        g.write(f'"{k}",{v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0):.0f},{v.get("SQ_BUSY_CU_CYCLES", 0):.0f},{gui:.0f},{frac:.3f}\n')
print(open(f"{out}/{tag}_mfma_pmc.csv").read())
PY
rm -rf $OUT/${TAG}_pmcm_*
