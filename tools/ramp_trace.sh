#!/bin/bash
# Per-launch duration of the K7 kernel in dispatch order over a short bench.py run (rocprofv3 --kernel-trace): is the ramp at the start of
# the timed region in the kernels themselves or between them?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ramp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-baseline-updates 0 --no-parity --shard-envs-per-gpu 0 --no-probe > $O/out.json 2> $O/err.txt || { echo "bench.py under rocprofv3 failed or timed out (rc $?)" >&2; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$O/tr/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k7 = [r for r in rows if "k_mlp_step3" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in k7]
gap = [(int(k7[i + 1]["Start_Timestamp"]) - int(k7[i]["End_Timestamp"])) / 1e3 for i in range(len(k7) - 1)]
print(len(d), "K7 launches")
for u in range(0, len(d) // 16):
    dd = d[16 * u:16 * u + 16]
    gg = gap[16 * u:16 * u + 15]
    print(f"update {u:2d}: K7 mean {sum(dd) / len(dd):7.2f} us (min {min(dd):7.2f}), gap between K7s mean {sum(gg) / max(len(gg), 1):6.2f} us")
PY
rm -rf $O/tr
