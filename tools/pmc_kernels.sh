#!/bin/bash
# SQ / TCC counters of named kernels (what do the waves wait on?), one rocprofv3 --pmc pass per counter set.  Run on the GPU box from the repo root.
# usage: bash tools/pmc_kernels.sh <tag> "<kernel substring> [<kernel substring> ...]" [extra bench.py flags]   (writes gpurun_out/<tag>_pmc_kernels.txt)
set -u
set -o pipefail
TAG=${1:?usage: pmc_kernels.sh <tag> "<kernel substrings>" [bench flags]}
PATS=${2:?kernel name substrings}
shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
n_ok=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_k_$t" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-standin --no-config3 --no-config4 --images 38 --no-steady-state "$@" > /dev/null 2>&1 && n_ok=$((n_ok + 1))
done
cd "$ROOT"
if [ $n_ok -eq 0 ]; then echo "pmc_kernels.sh: every rocprofv3 pass failed" >&2; exit 1; fi
PATS="$PATS" TAG="$TAG" python - <<'PY'
import collections, csv, glob, os, sys
pats = os.environ["PATS"].split()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = glob.glob("gpurun_out/pmc_k_*/**/*counter_collection.csv", recursive=True)
if not files:
    sys.exit("pmc_kernels.sh: no counter_collection.csv found")
for f in files:
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Kernel_Name"]:
                acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = f"gpurun_out/{os.environ['TAG']}_pmc_kernels.txt"
with open(out, "w") as g:
    for key, d in acc.items():
        g.write(f"== {key}\n")
        for c, v in sorted(d.items()):
            g.write(f"  {c:45s} mean/launch {sum(v)/len(v):16.1f}   launches {len(v)}\n")
print(open(out).read())
PY
rm -rf "$OUT"/pmc_k_*
