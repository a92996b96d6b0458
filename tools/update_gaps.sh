#!/bin/bash
# gaps between the kernels of the update's main chain (K7 -> reduce -> adam-chain -> K7 ...) from a rocprofv3 kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gap -- python3 $R/bench.py --cpu-baseline-updates 0 --steps 20 --warmup 3 --no-probe > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, statistics as st
f = glob.glob("gpurun_out/trace_gap/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("k_mlp_step", "k_mlp_reduce", "k_adam_chain", "k_adv_stats", "k_gae", "k_pack_rec64", "k_mlp_act"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("k_mlp_step", "k_mlp_reduce", "k_adam_chain", "k_adv_stats", "k_gae", "k_pack_rec64", "k_mlp_act"):
        if k in n: return k
gaps = {}
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 200: gaps.setdefault((short(a["Kernel_Name"]), short(b["Kernel_Name"])), []).append(g)
dur = {}
for r in rows:
    dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    print(f"{k:14s} n={len(v):4d} duration median {st.median(v):7.2f} us  min {min(v):7.2f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 10: print(f"{k[0]:14s} -> {k[1]:14s} n={len(v):4d} gap median {st.median(v):6.2f} us  mean {st.mean(v):6.2f} us")
PY
rm -rf gpurun_out/trace_gap
