#!/bin/bash
# One training step as the GPU saw it: rocprofv3 --kernel-trace of a short bench run -> tools/timeline.py.  Run on the GPU box from the repo root.
# usage: bash tools/timeline.sh <tag> [extra bench.py flags]     (writes gpurun_out/<tag>_timeline.txt)
set -u
set -o pipefail
TAG=${1:?usage: timeline.sh <tag> [bench flags]}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_tl" -- python3 "$ROOT/bench.py" --steps 30 --warmup 10 --no-cpu-baseline --no-standin --no-config3 --no-config4 --no-steady-state "$@" > "$OUT/${TAG}_timeline_run.log" 2>&1
rc=$?
cd "$ROOT"
f=$(find "$OUT/prof_tl" -name "*kernel_trace.csv" | head -1)
if [ $rc -ne 0 ] || [ -z "$f" ]; then
  echo "timeline.sh: rocprofv3 failed (exit $rc) or wrote no kernel trace; see $OUT/${TAG}_timeline_run.log" >&2
  exit 1
fi
python tools/timeline.py "$f" 30 3 > "$OUT/${TAG}_timeline.txt"
rm -rf "$OUT/prof_tl"
head -3 "$OUT/${TAG}_timeline.txt"
