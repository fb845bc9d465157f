# round-2 side measurements: proposal scatter variants (serial breakdown + bench), kernel statistics of the fused full-NeRFPlayer trainer
set -x
python -m pytest tests/test_gpu_kplanes.py tests/test_gpu_determinism.py tests/test_gpu_trainer.py -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --breakdown --no-overlap --steps 60 2>&1 >/dev/null | grep -E "prop|sum" 
python bench.py --no-cpu-baseline --steps 100 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'])"
if [ "$1" = "np" ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_npfull -- python3 $GRAFT_REPO_ROOT/tools/train_psnr_nerfplayer_full.py --steps 300 --width 240 --out $GRAFT_REPO_ROOT/gpurun_out/npfull_short.json > $GRAFT_REPO_ROOT/gpurun_out/npfull_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_npfull -name "*kernel_stats.csv" | head -1)
head -45 "$f" > gpurun_out/r02_npfull_kernel_stats.csv
rm -rf gpurun_out/prof_npfull
fi
