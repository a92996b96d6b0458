#!/bin/bash
# every kernel between the last k_adam_chain of one update and the first K7 launch of the next (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_head -- python3 $R/bench.py --cpu-baseline-updates 0 --steps 6 --warmup 3 --no-probe > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/trace_head/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
for g in glob.glob("gpurun_out/trace_head/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", ""), ""))
rows.sort()
# last update in the trace: from the last k_mlp_act back to the preceding k_adam_chain, forward to the 2nd k_mlp_step
acts = [i for i, r in enumerate(rows) if "k_mlp_act" in r[2]]
i0 = acts[-1]
while i0 > 0 and "k_adam_chain" not in rows[i0][2]: i0 -= 1
t0 = rows[i0][1]
n7 = 0
for r in rows[i0:]:
    print(f"{(r[0] - t0) / 1e3:9.2f} .. {(r[1] - t0) / 1e3:9.2f} us  q={r[3]:3s} {r[2]}")
    if "k_mlp_step" in r[2]:
        n7 += 1
        if n7 == 2: break
PY
rm -rf gpurun_out/trace_head
