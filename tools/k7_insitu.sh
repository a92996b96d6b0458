#!/bin/bash
# in-situ K7 durations (inside the update graph) from a rocprofv3 kernel trace of bench.py; env knobs inherited
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_k7 -- python3 $R/bench.py --cpu-baseline-updates 0 "$@" > /tmp/k7_bench.json 2>/dev/null
cd $R
python3 - <<PY
import csv, glob, json, statistics as st
f = glob.glob("gpurun_out/trace_k7/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def dur(key):
    return sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"])
d = dur("k_mlp_step")
msg = "K7 n=%d mean=%.1f median=%.1f p90=%.1f max=%.1f" % (len(d), st.mean(d), st.median(d), d[int(0.9 * len(d))], d[-1])
for k in ("k_mt_fill", "k_fy_accept", "k_fy_link", "k_fy_resolve", "k_mlp_reduce", "k_adv_stats"):
    x = [v for v in dur(k) if v > 3]
    if x: msg += " | %s mean=%.1f p90=%.1f" % (k, st.mean(x), x[int(0.9 * len(x))])
print(msg + " | ms_per_step=%.3f" % json.load(open("/tmp/k7_bench.json"))["ms_per_step"])
PY
rm -rf gpurun_out/trace_k7
