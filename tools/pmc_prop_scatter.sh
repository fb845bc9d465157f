# SQ counters of the proposal scatter (kplanes_gather_bwd_kernel<8,6>) and, for comparison, the field's grouped scatter: what do the waves wait on?
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"; do
  tag=$(echo $SET | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_ps_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --images 38 --no-steady-state > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ps_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        key = "prop_scatter" if "kplanes_gather_bwd_kernel<8" in k else ("field_scatter" if "scatter_grouped_kernel" in k else ("gradvec" if "gradvec_kernel" in k else None))
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r02_pmc_prop_scatter.txt", "w") as g:
    for key, d in acc.items():
        g.write(f"== {key}\n")
        for c, v in sorted(d.items()):
            g.write(f"  {c:45s} mean/launch {sum(v)/len(v):16.1f}   launches {len(v)}\n")
print(open("gpurun_out/r02_pmc_prop_scatter.txt").read())
PY
rm -rf gpurun_out/pmc_ps_*
