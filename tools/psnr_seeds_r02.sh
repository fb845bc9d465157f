# final-code PSNR statistics of round 2: 5 seeds each of fp32, bf16 (default: fused forward) and bf16 + bf16 gradient vectors
set -x
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5 --mlp-operands bf16 --eval-every 30000 --out gpurun_out/psnr_r02s_bf16.json > gpurun_out/psnr_r02s_bf16.log 2>&1
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5 --mlp-operands fp32 --eval-every 30000 --out gpurun_out/psnr_r02s_fp32.json > gpurun_out/psnr_r02s_fp32.log 2>&1
python tools/train_psnr.py --steps 30000 --seeds 1,2,3,4,5 --mlp-operands bf16 --gvec-dtype bf16 --eval-every 30000 --out gpurun_out/psnr_r02s_bf16_gvec16.json > gpurun_out/psnr_r02s_bf16_gvec16.log 2>&1
grep -h "==" gpurun_out/psnr_r02s_*.log
