"""Print the top rows of a rocprofv3 kernel_stats.csv (dev tool). usage: python tools/kstats.py <dir> <steps> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 22
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} ms/step {float(r['TotalDurationNs'])/1e6/steps:7.3f}")
