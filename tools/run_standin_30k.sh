#!/bin/bash
# One 30 000-step run of the reference-algorithm stand-in (stock PyTorch, channel-last planes gathered as rows: ~4 % faster than F.grid_sample, which is
# what lets 30 000 steps + scene rendering + evaluation fit the 3 600 s of one GPU call).  usage: bash tools/run_standin_30k.sh <scene> <seed>
set -u
SCENE=${1:-default}; SEED=${2:-3}
OUT=gpurun_out; mkdir -p $OUT
python tools/train_psnr.py --standin --standin-layout hwc --steps 30000 --seeds $SEED --eval-frames 8 --eval-every 30000 --scene $SCENE --train-budget-s 3410 \
  --out $OUT/r05_psnr_30k_standin_${SCENE}_seed${SEED}.json > $OUT/r05_psnr_30k_standin_${SCENE}_seed${SEED}.log 2>&1
tail -n 3 $OUT/r05_psnr_30k_standin_${SCENE}_seed${SEED}.log
