#!/bin/bash
# Round-5 PSNR@30k runs of the HIP trainers on current code (run on the GPU box from the repo root):
#   k-planes preset, default + textured scene, seeds 1-3 each (tools/train_psnr.py), and the nerfplayer-nerfacto preset on the stadium-players
#   scene (tools/train_psnr_nerfplayer.py --scene stadium).  Outputs: gpurun_out/r05_psnr_*.json (copied to profiles/ by hand).
set -u
set -o pipefail
OUT=gpurun_out
mkdir -p $OUT
python tools/train_psnr.py --steps 30000 --seeds 1,2,3 --eval-frames 8 --out $OUT/r05_psnr_30k_bf16.json > $OUT/r05_psnr_30k_bf16.log 2>&1 || echo "default-scene run failed" >&2
python tools/train_psnr.py --steps 30000 --seeds 1,2,3 --eval-frames 8 --scene textured --out $OUT/r05_psnr_30k_bf16_textured.json > $OUT/r05_psnr_30k_bf16_textured.log 2>&1 || echo "textured-scene run failed" >&2
python tools/train_psnr_nerfplayer.py --steps 30000 --scene stadium --width 960 --frames 100 --out $OUT/r05_psnr_nerfplayer_stadium_30k.json > $OUT/r05_psnr_nerfplayer_stadium_30k.log 2>&1 || echo "nerfplayer stadium run failed" >&2
for f in $OUT/r05_psnr_30k_bf16.log $OUT/r05_psnr_30k_bf16_textured.log $OUT/r05_psnr_nerfplayer_stadium_30k.log; do tail -n 2 $f; done
