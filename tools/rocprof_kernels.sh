#!/bin/bash
# usage: tools/rocprof_kernels.sh <tag> <bench_kernels args...>   (env AURPPO_* knobs are inherited)
# Runs tools/bench_kernels.py under rocprofv3 --kernel-trace --stats and prints our kernels' averages.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kp_$tag -- python3 $R/tools/bench_kernels.py "$@" > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/kp_$tag/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_gather", "k_gae", "k_loss", "k_adv", "k_fy", "k_mt_", "k_mlp", "k_clip", "k_sqnorm", "index_select", "copyBuffer")):
        print(f"$tag {n[:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
rm -rf gpurun_out/kp_$tag
