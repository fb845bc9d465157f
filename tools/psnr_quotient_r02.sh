# PSNR@30k of the round's final default (fused forward + quotient scatter): 5 seeds bf16, 5 seeds fp32 -> profiles/r02_psnr_30k_{bf16,fp32}.json
# (the fp32 file of the round was merged from two invocations, seeds 1-3 and 4-5, on the same code)
set -x
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5 --mlp-operands bf16 --eval-every 30000 --out gpurun_out/psnr_r02q_bf16.json > gpurun_out/psnr_r02q_bf16.log 2>&1
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5 --mlp-operands fp32 --eval-every 30000 --out gpurun_out/psnr_r02q_fp32.json > gpurun_out/psnr_r02q_fp32.log 2>&1
grep -h "==" gpurun_out/psnr_r02q_*.log
