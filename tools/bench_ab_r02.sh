# A-B of the fused field kernels (bench.py, bf16 operands): run on the GPU box from the repo root
set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02_t4.log
B="python bench.py --no-cpu-baseline --steps 100 --warmup 10"
$B --fused-field > gpurun_out/r02_bench_fused.json 2> gpurun_out/r02_bench_fused.err
$B --fused-field --fused-forward-only > gpurun_out/r02_bench_fusedfwd.json 2>> gpurun_out/r02_bench_fused.err
$B > gpurun_out/r02_bench_unfused.json 2>> gpurun_out/r02_bench_fused.err
$B --no-steady-state --breakdown --no-overlap --steps 40 2> gpurun_out/r02_breakdown_fused.txt > /dev/null
$B --no-steady-state --breakdown --no-overlap --steps 40 --no-fused-field 2> gpurun_out/r02_breakdown_unfused.txt > /dev/null
$B --no-steady-state --breakdown --no-overlap --steps 40 --no-fused-backward 2> gpurun_out/r02_breakdown_fusedfwd.txt > /dev/null
python - <<'PY'
import json
for n in ("fused","fusedfwd","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/r02_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],3), round(d["value"]), "steady", round(d["steady_state"]["ms_per_step"],3), round(d["steady_state"]["value"]))
    except Exception as e: print(n, "ERR", e)
PY
