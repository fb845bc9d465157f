# one training step as the GPU saw it: rocprofv3 --kernel-trace of a short bench run -> tools/timeline.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-steady-state $@ > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py "$f" 30 3 > gpurun_out/r02_timeline.txt
rm -rf gpurun_out/prof_tl
head -3 gpurun_out/r02_timeline.txt
