#!/bin/bash
# per-kernel durations of the 4-epoch shuffle pipeline (tools/bench_shuffle_pipe.py) from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_k2 -- python3 $R/tools/bench_shuffle_pipe.py
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/trace_k2/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_fy", "k_mt_", "fillBuffer")):
        print(f"{n[:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf gpurun_out/trace_k2
