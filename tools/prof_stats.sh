#!/bin/bash
# usage: tools/prof_stats.sh <tag> <python script + args...>
# rocprofv3 --kernel-trace --stats of `python3 <script> <args>`; prints our kernels' averages and keeps the CSV
# as gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_$tag -- python3 $R/"$@" > $R/gpurun_out/$tag.stdout 2> $R/gpurun_out/$tag.stderr
cd $R
f=$(ls gpurun_out/ps_$tag/*/*kernel_stats.csv | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")):
    n = r["Name"]
    if any(k in n for k in ("k_gather", "k_gae", "k_loss", "k_adv", "k_fy", "k_mt_", "k_mlp", "k_clip", "k_sqnorm", "k_adam", "k_pack", "fillBuffer", "copyBuffer", "rccl", "Rccl", "nccl")):
        print(f"$tag {n[:64]:64s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} min_us={float(r['MinNs'])/1e3:9.2f} max_us={float(r['MaxNs'])/1e3:9.2f}")
PY
tail -n 3 gpurun_out/$tag.stdout
rm -rf gpurun_out/ps_$tag
