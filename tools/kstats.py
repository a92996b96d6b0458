"""Print a rocprofv3 kernel_stats.csv as 'calls  avg_us  name' (kernel names contain commas).  python tools/kstats.py <dir or csv> [top]"""
import csv, glob, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(p + "/**/*kernel_stats.csv", recursive=True))[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:top]:
    print(f"{int(r['Calls']):6d}  {float(r['AverageNs']) / 1e3:9.1f} us  (min {float(r['MinNs']) / 1e3:8.1f}, max {float(r['MaxNs']) / 1e3:8.1f})  {r['Name'][:90]}")
