# round-2 A-B runs of bench.py on the GPU box (from the repo root)
set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02_t5.log
B="python bench.py --no-cpu-baseline --steps 100 --warmup 10"
for v in "" "--prop-after-field" "--no-overlap" "--mlp-operands fp32" "--gvec-dtype-bf16"; do
  n=$(echo "default$v" | tr -d ' -')
  if [ "$v" = "--gvec-dtype-bf16" ]; then continue; fi
  $B $v > gpurun_out/r02_bench_$n.json 2> gpurun_out/r02_bench_$n.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_bench_default*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("r02_bench_")[1], round(d["ms_per_step"],3), round(d["value"]), "steady", round(d["steady_state"]["ms_per_step"],3), round(d["steady_state"]["value"]), d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],3))
    except Exception as e: print(f, "ERR", e)
PY
bash tools/collect_profiles.sh r02
