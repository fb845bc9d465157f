#!/bin/bash
# Run on the GPU box from the repo root: config 4 (nerfplayer-nerfacto preset, fused trainer) -- bench line, rocprofv3 kernel stats, FETCH / WRITE
# counters of its kernels, and a `roofline` object in BOTH byte conventions of SURVEY 8d.
# usage: bash tools/collect_config4_profiles.sh <tag>    -> gpurun_out/<tag>_nerfplayer_fused_{bench.json,kernel_stats.csv,pmc.csv}
set -u
set -o pipefail
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
fail() { echo "collect_config4_profiles.sh: $1" >&2; exit 1; }
STEPS=30; WARM=5
NPFLAGS=${NPFLAGS---tiled}   # round 6: the main table through the owner-computes pass (csrc/tgrid_tiles.hip); NPFLAGS= for the round-5 form
python tools/bench_nerfplayer.py --fused --stadium $NPFLAGS 2> $OUT/${TAG}_np_bench.err | tail -1 > $OUT/${TAG}_np_bench_line.json || fail "bench_nerfplayer.py --fused --stadium failed"
# the round-4 workload (random rays through the box) once more, for the comparison with profiles/r04_nerfplayer_fused_bench.json
python tools/bench_nerfplayer.py --fused $NPFLAGS 2> $OUT/${TAG}_np_random_rays_bench.err | tail -n 1 > $OUT/${TAG}_nerfplayer_fused_random_rays_bench.json \
  || fail "bench_nerfplayer.py --fused (random rays) failed (see $OUT/${TAG}_np_random_rays_bench.err)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_np_stats -- python3 $ROOT/tools/bench_nerfplayer.py --fused --stadium $NPFLAGS --steps $STEPS --warmup $WARM > $OUT/${TAG}_np_stats.log 2>&1 \
  || fail "rocprofv3 --stats pass failed (see $OUT/${TAG}_np_stats.log)"
find $OUT/${TAG}_np_stats -name '*kernel_stats.csv' | grep -q . || fail "no kernel_stats.csv"
# three consecutive steps as the GPU saw them (queues, start offsets): tools/timeline.py on the same trace
TRACE=$(find $OUT/${TAG}_np_stats -name '*kernel_trace.csv' | head -1)
[ -n "$TRACE" ] && python $ROOT/tools/timeline.py "$TRACE" 10 3 > $OUT/${TAG}_nerfplayer_fused_timeline.txt 2> $OUT/${TAG}_np_timeline.err || echo "collect_config4_profiles.sh: no timeline (see $OUT/${TAG}_np_timeline.err)" >&2
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_np_pmc_$C -- python3 $ROOT/tools/bench_nerfplayer.py --fused --stadium $NPFLAGS --steps 6 --warmup 2 > $OUT/${TAG}_np_pmc_$C.log 2>&1 \
    || fail "rocprofv3 --pmc $C pass failed (see $OUT/${TAG}_np_pmc_$C.log)"
  find $OUT/${TAG}_np_pmc_$C -name '*counter_collection.csv' | grep -q . || fail "no counter_collection.csv from the --pmc $C pass"
done
cd $ROOT
TAG=$TAG STEPS=$((STEPS + WARM)) NPFLAGS="$NPFLAGS" python - <<'PY'
import collections, csv, glob, json, os
tag, out, steps = os.environ["TAG"], "gpurun_out", int(os.environ["STEPS"])
rows = list(csv.DictReader(open(glob.glob(f"{out}/{tag}_np_stats/**/*kernel_stats.csv", recursive=True)[0])))
with open(f"{out}/{tag}_nerfplayer_fused_kernel_stats.csv", "w") as g:
    g.write(f"# rocprofv3 --kernel-trace --stats -- python3 tools/bench_nerfplayer.py --fused --stadium {os.environ.get('NPFLAGS', '')} --steps 30 --warmup 5   ({steps} steps in the trace)\n")
    w = csv.writer(g)
    w.writerow(rows[0].keys())
    for r in rows[:30]:
        w.writerow(r.values())
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{tag}_np_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                pmc[r["Kernel_Name"].split("(")[0].replace("void ", "")][c].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_nerfplayer_fused_pmc.csv", "w") as g:
    g.write("# rocprofv3 --pmc <C> --kernel-trace -- python3 tools/bench_nerfplayer.py --fused --stadium --steps 6 --warmup 2 (one counter per pass); SUM over the launches of one step\n")
    g.write("# KiB as reported; traffic = 2 x FETCH + WRITE (MI355X_MICROARCH.md's gfx950 correction; calibrated for 16-B-per-lane streams = the Adam kernels, a ratio-only figure for the scattered tgrid accesses)\n")
    g.write("kernel,launches_per_step,FETCH_SIZE_KiB_per_step,WRITE_SIZE_KiB_per_step,traffic_bytes_per_step\n")
    per_step = {}
    for k, d in sorted(pmc.items()):
        n = max(len(d.get("FETCH_SIZE", [])), len(d.get("WRITE_SIZE", [])))
        fe, wr = sum(d.get("FETCH_SIZE", [])) / 8, sum(d.get("WRITE_SIZE", [])) / 8  # 8 steps in a counter pass
        per_step[k] = (2 * fe + wr) * 1024
        if "snerf" in k:
            g.write(f'"{k}",{n / 8:.2f},{fe:.1f},{wr:.1f},{per_step[k]:.0f}\n')
line = json.load(open(f"{out}/{tag}_np_bench_line.json"))
R, P = line["rays"], line["params"]
ms = lambda pat: sum(float(r["TotalDurationNs"]) for r in rows if pat in r["Name"]) / steps / 1e6
tr = lambda pat: sum(v for k, v in per_step.items() if pat in k)
adam_ms = ms("adam_tv_kernel") + ms("adam_kernel")
tiles_ms, bin_ms = ms("tt_tiles_kernel"), ms("tt_bin_kernel") + ms("tt_scan_") + ms("tt_positions_kernel")
fwd_ms, bwd_ms = ms("tgrid_kernel<false") + ms("tgrid_fwd_runs_kernel"), ms("tgrid_kernel<true") + ms("tgrid_bwd_runs_kernel")
alg_ray, sec_ray = 242688, 256 * 40 * 64 + 96 * 40 * 64 + 48 * 128 * 64  # algorithmic bytes / 64-B sectors touched per ray, forward
line["roofline"] = {
    "bound": "hbm", "kernel": "optimiser sweep: adam_tv_kernel (temporal grids, TV term fused) + adam_kernel (MLPs): p, g, m, v read + p, m, v written + g cleared = 32 B / parameter",
    "achieved": 32 * P / (adam_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": 32 * P / (adam_ms * 1e-3) / 1e9 / 8000.0,
    "algorithmic_per_step": 32 * P, "kernel_ms_per_step": adam_ms, "traffic": tr("adam_tv_kernel") + tr("adam_kernel"),
    "traffic_note": "PMC bytes per step (2 x FETCH_SIZE + WRITE_SIZE, separate passes); below the algorithmic figure because gradient quads that are zero already are not cleared and untouched rows' gradient reads hit zero lines",
    "tgrid_kernel": {
        "forward_ms_per_step": fwd_ms, "backward_ms_per_step": bwd_ms,
        "algorithmic_bytes_per_step_forward": alg_ray * R, "sector_granular_bytes_per_step_forward": sec_ray * R,
        "forward_algorithmic_GBps": alg_ray * R / (fwd_ms * 1e-3) / 1e9, "forward_sector_granular_GBps": sec_ray * R / (fwd_ms * 1e-3) / 1e9,
        "backward_algorithmic_GBps_rmw": 2 * alg_ray * R / (bwd_ms * 1e-3) / 1e9, "backward_sector_granular_GBps_rmw": 2 * sec_ray * R / (bwd_ms * 1e-3) / 1e9,
        "pmc_traffic_bytes_per_step": {"forward": tr("tgrid_kernel<false") + tr("tgrid_fwd_runs_kernel"), "backward": tr("tgrid_kernel<true") + tr("tgrid_bwd_runs_kernel")},
        "conventions": "SURVEY 8d: per sample 16 levels x 8 corners x 3 floats x 4 B = 1536 B algorithmic (main) / 480 B (proposal levels); each corner row is its own 64-B sector: "
                       "128 / 40 sectors per sample.  Per ray 256 x 480 + 96 x 480 + 48 x 1536 = 242 688 B algorithmic, 1.29 MB sector-granular; the backward reads and writes them (x 2)"},
    "source": f"rocprofv3 --kernel-trace --stats over {steps} steps (profiles/{tag}_nerfplayer_fused_kernel_stats.csv) and separate --pmc passes (profiles/{tag}_nerfplayer_fused_pmc.csv)"}
line["flags"] = os.environ.get("NPFLAGS", "")
if tiles_ms > 0:
    # round 6: the main table (6 119 864 rows x 66 floats) is stepped by tt_tiles_kernel: 24 B / parameter (p, m, v read and written; no dense gradient)
    n_main = 6119864 * 66
    line["roofline_tiles"] = {"bound": "hbm", "kernel": "tt_tiles_kernel<2,1>: gradient scatter (LDS, one owner per 256-row tile) + temporal TV + Adam of the main table in one pass",
                              "achieved": 24 * n_main / (tiles_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": 24 * n_main / (tiles_ms * 1e-3) / 1e9 / 8000.0,
                              "algorithmic_per_step": 24 * n_main, "kernel_ms_per_step": tiles_ms, "traffic": tr("tt_tiles_kernel"),
                              "binning_ms_per_step": bin_ms, "binning_traffic": tr("tt_bin_kernel") + tr("tt_scan_") + tr("tt_positions_kernel")}
json.dump(line, open(f"{out}/{tag}_nerfplayer_fused_bench.json", "w"), indent=1)
print(json.dumps(line)[:1500])
PY
rm -rf $OUT/${TAG}_np_stats $OUT/${TAG}_np_pmc_FETCH_SIZE $OUT/${TAG}_np_pmc_WRITE_SIZE $OUT/${TAG}_np_bench_line.json
ls -la $OUT | grep ${TAG}_nerfplayer_fused
