#!/bin/bash
# Round-5 PSNR@30k on the round's LAST code: k-planes preset, default and textured scene, seeds 1-10 each (run on the GPU box from the repo root).
set -u
set -o pipefail
OUT=gpurun_out
mkdir -p $OUT
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5,6,7,8,9,10 --eval-frames 8 --out $OUT/r05_psnr_30k_bf16_final_code.json > $OUT/r05_psnr_30k_bf16_final_code.log 2>&1 || echo "default-scene run failed" >&2
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5,6,7,8,9,10 --eval-frames 8 --scene textured --out $OUT/r05_psnr_30k_bf16_textured_final_code.json > $OUT/r05_psnr_30k_bf16_textured_final_code.log 2>&1 || echo "textured-scene run failed" >&2
for f in $OUT/r05_psnr_30k_bf16_final_code.log $OUT/r05_psnr_30k_bf16_textured_final_code.log; do tail -n 1 $f; done
