# A-B series of round 2 (tools/train_psnr.py, 30 k steps each): which 16-bit pieces cost novel-view PSNR.  Results: profiles/r02_psnr_ab.md
set -x
T="python tools/train_psnr.py --steps 30000 --no-fused-field --eval-every 30000"
$T --seeds 2,3 --mlp-operands bf16 --gvec-dtype fp32 --out gpurun_out/psnr_r02c_bf16_gv32.json > gpurun_out/psnr_r02c_bf16_gv32.log 2>&1
$T --seeds 3 --mlp-operands bf16 --gvec-dtype bf16 --out gpurun_out/psnr_r02c_bf16_gv16.json > gpurun_out/psnr_r02c_bf16_gv16.log 2>&1
$T --seeds 2,3 --mlp-operands fp16 --gvec-dtype bf16 --out gpurun_out/psnr_r02c_fp16_gv16.json > gpurun_out/psnr_r02c_fp16_gv16.log 2>&1
$T --seeds 3 --mlp-operands fp32 --out gpurun_out/psnr_r02c_fp32.json > gpurun_out/psnr_r02c_fp32.log 2>&1
grep -h "==" gpurun_out/psnr_r02c_*.log
