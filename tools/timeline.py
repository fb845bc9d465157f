"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (dev tool).
usage: python tools/timeline.py <kernel_trace.csv> [step_index_from_end] [n_steps] [marker kernel substring (default raygen_kernel)]"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # consecutive steps to print (the optimiser sweep of step k runs under the head of step k + 1)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("snerf::", "").replace("void ", "")[:60]
# a step starts at raygen_kernel (the NeRFPlayer tools feed random rays: their steps start at spaced_bins_kernel)
marker = sys.argv[4] if len(sys.argv) > 4 else "raygen_kernel"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
i0, i1 = starts[-back - 1], starts[-back - 1 + nsteps]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in step)
print(f"step wall (first start -> last end): {(end - t0) / 1e6:.3f} ms, {len(step)} kernels")
busy = []
for r in step:
    busy.append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0))
busy.sort()
cov, cur_s, cur_e = 0, None, None
for s, e in busy:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            cov += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print(f"GPU busy (union of kernel intervals): {cov / 1e6:.3f} ms; sum of kernel durations {sum(e - s for s, e in busy) / 1e6:.3f} ms")
for r in step:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3} {name(r)}")
