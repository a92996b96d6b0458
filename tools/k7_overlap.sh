#!/bin/bash
# K7's in-situ duration grouped by which side-stream kernels overlapped the launch (rocprofv3 kernel trace of bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_ov -- python3 $R/bench.py --cpu-baseline-updates 0 --steps 30 --warmup 3 --no-probe > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, statistics as st
f = glob.glob("gpurun_out/trace_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
side = [(s, e, k) for s, e, n in rows for k in ("k_mt_fill", "k_fy_accept", "k_fy_link", "k_fy_resolve", "fillBuffer") if k in n]
groups = {}
for s, e, n in rows:
    if "k_mlp_step" not in n: continue
    ov = {}
    for a, b, k in side:
        o = min(e, b) - max(s, a)
        if o > 0: ov[k] = ov.get(k, 0) + o
    key = "+".join(sorted(k for k, o in ov.items() if o > 0.05 * (e - s))) or "nothing"
    groups.setdefault(key, []).append((e - s) / 1e3)
for k, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(v):4d} launches overlapped by {k:55s} median {st.median(v):7.1f} us  mean {st.mean(v):7.1f}  min {min(v):7.1f}")
PY
rm -rf gpurun_out/trace_ov
