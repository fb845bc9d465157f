#!/bin/bash
# SQ counters of named kernels while a TOOL runs (not bench.py): one rocprofv3 --pmc pass per counter set.  Run on the GPU box from the repo root.
# usage: bash tools/pmc_tool.sh <tag> "<kernel substring> [...]" tools/<script>.py [script args]      (writes gpurun_out/<tag>_pmc_tool.txt)
set -u
set -o pipefail
TAG=${1:?usage: pmc_tool.sh <tag> "<kernel substrings>" <script> [args]}
PATS=${2:?kernel name substrings}
SCRIPT=${3:?python script}
shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
n_ok=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA"; do
  t=$(echo $SET | tr ' ' '_' | cut -c1-40)
  if rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_t_$t" -- python3 "$ROOT/$SCRIPT" "$@" > "$OUT/pmc_t_$t.log" 2>&1; then n_ok=$((n_ok + 1)); else echo "pass '$SET' failed" >&2; fi
done
cd "$ROOT"
if [ $n_ok -eq 0 ]; then echo "pmc_tool.sh: every rocprofv3 pass failed" >&2; exit 1; fi
PATS="$PATS" TAG="$TAG" python - <<'PY'
import collections, csv, glob, os, sys
pats = os.environ["PATS"].split()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
files = glob.glob("gpurun_out/pmc_t_*/**/*counter_collection.csv", recursive=True)
if not files:
    sys.exit("pmc_tool.sh: no counter_collection.csv found")
for f in files:
    for r in csv.DictReader(open(f)):
        for p in pats:
            if p in r["Kernel_Name"]:
                acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = f"gpurun_out/{os.environ['TAG']}_pmc_tool.txt"
with open(out, "w") as g:
    for key, d in acc.items():
        g.write(f"== {key}\n")
        for c, v in sorted(d.items()):
            g.write(f"  {c:45s} mean/launch {sum(v)/len(v):16.1f}   launches {len(v)}\n")
print(open(out).read())
PY
rm -rf "$OUT"/pmc_t_*
