# final-code PSNR runs of round 2 (default configuration: bf16 MLP operands, fp32 gradient vectors), 3 seeds; plus profile collection
set -x
python tools/train_psnr.py --steps 30000 --seeds 1,2,3 --mlp-operands bf16 --eval-every 30000 --out gpurun_out/psnr_r02f_bf16.json > gpurun_out/psnr_r02f_bf16.log 2>&1
grep -h "==" gpurun_out/psnr_r02f_bf16.log
bash tools/collect_profiles.sh r02
