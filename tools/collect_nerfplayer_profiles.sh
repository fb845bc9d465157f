#!/bin/bash
# Run on the GPU box from the repo root: rocprofv3 kernel stats of the NeRFPlayer configurations (config 4 fused trainer, config 4 through
# the nerfstudio-shaped model, full NeRFPlayer).  usage: bash tools/collect_nerfplayer_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
set -u
set -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
for MODE in fused model full; do
  FLAG=""; [ $MODE = fused ] && FLAG="--fused"; [ $MODE = full ] && FLAG="--full"
  python tools/bench_nerfplayer.py $FLAG 2>/dev/null | tail -1 > $OUT/${TAG}_nerfplayer_${MODE}_bench.json
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_np_$MODE -- python3 $ROOT/tools/bench_nerfplayer.py $FLAG --steps 30 --warmup 5 > /dev/null 2>&1 )
  python - <<PY
import csv, glob
f = glob.glob("$OUT/${TAG}_np_$MODE/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.reader(open(f[0])))
    with open("$OUT/${TAG}_nerfplayer_${MODE}_kernel_stats.csv", "w") as g:
        g.write("# rocprofv3 --kernel-trace --stats -- python3 tools/bench_nerfplayer.py $FLAG --steps 30 --warmup 5   (35 steps in the trace)\n")
        csv.writer(g).writerows(rows[:30])
PY
  rm -rf $OUT/${TAG}_np_$MODE
done
ls -la $OUT | grep ${TAG}_nerfplayer
