#!/bin/bash
# concurrency probe: 1 vs 3 stand-in processes, 200 steps each
mkdir -p gpurun_out/conc
( time python tools/train_psnr.py --standin --standin-layout hwc --steps 200 --seeds 1 --eval-frames 1 --eval-every 100000 --scene textured --out gpurun_out/conc/solo.json > gpurun_out/conc/solo.log 2>&1 ) 2> gpurun_out/conc/solo.time
for S in 1 2 3; do
( time python tools/train_psnr.py --standin --standin-layout hwc --steps 200 --seeds $S --eval-frames 1 --eval-every 100000 --scene textured --out gpurun_out/conc/c$S.json > gpurun_out/conc/c$S.log 2>&1 ) 2> gpurun_out/conc/c$S.time &
done
wait
grep -h "rays/s\|real" gpurun_out/conc/*.log gpurun_out/conc/*.time | tail -30
nproc
