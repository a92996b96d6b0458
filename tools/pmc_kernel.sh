#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-name-substring> <script + args>
# HBM bytes per launch of one kernel: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md section HBM:
# counters in KiB, FETCH_SIZE x2 on gfx950); prints and stores gpurun_out/<tag>_pmc.json
tag=$1; kern=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/"$@" > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/"$@" > /dev/null 2> $O/write.err
cd $R
python3 - <<PY
import csv, glob, json, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("$O/fetch", "$O/write"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "$kern" in r["Kernel_Name"]:
                import re
                m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
                key = (m.group(1) if m else r["Kernel_Name"][:40]) + " grid=" + r.get("Grid_Size", "?")
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in vals.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024
    w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024
    out[k] = {"launches": len(v["FETCH_SIZE"]), "fetch_raw_bytes": f, "fetch_corrected_bytes": 2 * f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w}
json.dump(out, open("$R/gpurun_out/${tag}_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O
