# round-2 side measurements: 6-scale field with bf16 operands; kernel statistics of the fused full-NeRFPlayer trainer
set -x
python -m pytest tests/test_gpu_mlp.py tests/test_gpu_nerfplayer_full_trainer.py -q 2>&1 | tail -5
python tools/bench_config3.py bf16 2>&1 | tail -1 > gpurun_out/r02_config3_bf16.json
python tools/bench_config3.py fp32 2>&1 | tail -1 > gpurun_out/r02_config3_fp32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_npfull -o npfull -- python3 $GRAFT_REPO_ROOT/tools/train_psnr_nerfplayer_full.py --steps 300 --width 240 --out $GRAFT_REPO_ROOT/gpurun_out/npfull_short.json > $GRAFT_REPO_ROOT/gpurun_out/npfull_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_npfull -name "*kernel_stats.csv" | head -1)
head -40 "$f" > gpurun_out/r02_npfull_kernel_stats.csv
rm -rf gpurun_out/prof_npfull
cat gpurun_out/r02_config3_bf16.json gpurun_out/r02_config3_fp32.json
