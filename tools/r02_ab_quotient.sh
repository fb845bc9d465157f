# A/B of the quotient form of the field scatter (default) against the product form (--no-quotient-scatter): serial breakdown + step times
python bench.py --no-cpu-baseline --breakdown --no-overlap --steps 40 2>&1 >/dev/null | head -12
for v in "" "--no-quotient-scatter" "" "--no-quotient-scatter"; do echo "== $v"; python bench.py --no-cpu-baseline --steps 100 --warmup 10 $v | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'steady', d['steady_state']['ms_per_step'])"; done
