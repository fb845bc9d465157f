set -x
python -m pytest tests/test_gpu_kplanes.py tests/test_gpu_field_fused.py tests/test_gpu_determinism.py tests/test_gpu_mlp.py tests/test_gpu_sharded.py -q 2>&1 | grep -E "passed|failed|Error" | tail -4
for v in "" "" ; do echo "== $v"; python bench.py --no-cpu-baseline --steps 100 --warmup 10 $v | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'])"; done
python bench.py --no-cpu-baseline --breakdown --no-overlap --steps 40 2>&1 >/dev/null | grep -E "sum|mlp_bwd|field_fwd"
