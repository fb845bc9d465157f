#!/bin/bash
# Several 30 000-step runs of the reference-algorithm stand-in (stock PyTorch; tools/run_standin_30k.sh) SIDE BY SIDE on one GPU: one stand-in leaves the
# MI355X mostly idle (thousands of small ATen kernels per step, ~0.1 s per step), so k processes share it at far less than k times the wall clock -- which is
# what makes n >= 4 seeds per scene affordable inside the round's GPU budget.  Each process has its own dataset copy, parameters and RNG streams: the runs are
# independent.  usage: bash tools/run_standin_30k_parallel.sh <scene> <tag> <seed> [<seed> ...]
set -u
SCENE=${1:-textured}; TAG=${2:-r06}; shift 2
OUT=gpurun_out; mkdir -p $OUT
PIDS=()
for SEED in "$@"; do
  python tools/train_psnr.py --standin --standin-layout hwc --steps 30000 --seeds $SEED --eval-frames 8 --eval-every 30000 --scene $SCENE --train-budget-s ${BUDGET_S:-5000} \
    --out $OUT/${TAG}_psnr_30k_standin_${SCENE}_seed${SEED}.json > $OUT/${TAG}_psnr_30k_standin_${SCENE}_seed${SEED}.log 2>&1 &
  PIDS+=($!)
done
RC=0
for P in "${PIDS[@]}"; do wait $P || RC=1; done
for SEED in "$@"; do tail -n 2 $OUT/${TAG}_psnr_30k_standin_${SCENE}_seed${SEED}.log; done
exit $RC
