# PSNR@30k with the quotient scatter (default): 5 seeds bf16 (gpurun_out/psnr_r02q_bf16.json) + 5 seeds fp32 (3 + 2)
set -x
python tools/train_psnr.py --steps 30000 --seeds 4,5 --mlp-operands fp32 --eval-every 30000 --out gpurun_out/psnr_r02q_fp32_45.json > gpurun_out/psnr_r02q_fp32_45.log 2>&1
grep -h "==" gpurun_out/psnr_r02q_fp32_45.log
