#!/bin/bash
# kernels of one rollout step (between two k_mlp_act launches) in train() on the synthetic env, from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_ro -- python3 $R/tools/bench_train.py 4096 1 16 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/trace_ro/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]) for r in csv.DictReader(open(f)))
acts = [i for i, r in enumerate(rows) if "k_mlp_act" in r[2]]
# a pair of consecutive act launches late in the trace, well inside a rollout
i0, i1 = acts[-40], acts[-39]
t0 = rows[i0][0]
for r in rows[i0:i1 + 1]:
    print(f"{(r[0] - t0) / 1e3:8.2f} .. {(r[1] - t0) / 1e3:8.2f} us  {r[2]}")
PY
rm -rf gpurun_out/trace_ro
