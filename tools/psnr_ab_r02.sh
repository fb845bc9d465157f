set -x
T="python tools/train_psnr.py --steps 30000 --seeds 1 --no-fused-field --eval-every 30000"
$T --mlp-operands fp16 --out gpurun_out/psnr_r02b_fp16.json > gpurun_out/psnr_r02b_fp16.log 2>&1
$T --mlp-operands bf16 --gvec-dtype fp32 --out gpurun_out/psnr_r02b_bf16_gv32.json > gpurun_out/psnr_r02b_bf16_gv32.log 2>&1
$T --mlp-operands fp32 --gvec-dtype bf16 --out gpurun_out/psnr_r02b_fp32_gv16.json > gpurun_out/psnr_r02b_fp32_gv16.log 2>&1
$T --mlp-operands fp32 --sigma-operands bf16 --out gpurun_out/psnr_r02b_sigma16.json > gpurun_out/psnr_r02b_sigma16.log 2>&1
$T --mlp-operands fp32 --color-operands bf16 --out gpurun_out/psnr_r02b_color16.json > gpurun_out/psnr_r02b_color16.log 2>&1
$T --mlp-operands fp32 --proposal-operands bf16 --out gpurun_out/psnr_r02b_prop16.json > gpurun_out/psnr_r02b_prop16.log 2>&1
D="python tools/train_psnr.py --steps 8000 --schedule-steps 8000 --seeds 1 --no-fused-field --eval-every 8000 --eval-frames 4 --deterministic --mlp-operands fp32"
$D --out gpurun_out/psnr_r02b_det_a.json > gpurun_out/psnr_r02b_det_a.log 2>&1
$D --out gpurun_out/psnr_r02b_det_b.json > gpurun_out/psnr_r02b_det_b.log 2>&1
grep -h "==" gpurun_out/psnr_r02b_*.log
