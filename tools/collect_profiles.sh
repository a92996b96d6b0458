#!/bin/bash
# Regenerates the round's committed evidence on the GPU box (outputs under gpurun_out/prof_final/; copy into profiles/rNN/):
#   bench.json                          python bench.py (default flags: parity gate + cpu_baseline + both rooflines)
#   bench_under_rocprof.json + bench_kernel_stats.csv            the same command under rocprofv3 --kernel-trace --stats
#   bench_nofused_*.json/csv            bench.py --no-fused-mlp (per-op path: K1, K3, K4+K5, K6b stand-alone durations)
#   bench_forcedp_*.json/csv            bench.py --force-dp (the W > 1 launch path captured around a one-rank RCCL all-reduce)
#   mlp_pmc.json                        separate --pmc passes for the K7 kernel's HBM traffic
#   bench_wide_{3x128,3x64,2x128}*.json/csv   bench.py --hidden-dim/--num-layers (K7w / K8w), plain and under rocprofv3
#   wide_bench.json                     tools/bench_wide.py: K7w / K8w against the per-op path over the -d / -nl shapes
# Each step runs under its own timeout; a step that times out stops the script (no GPU step after a hang).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_final
mkdir -p $O
cd $R
run() {  # run <seconds> <cmd...>
  local secs=$1; shift
  timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: $*" >&2; exit $rc; fi
  return $rc
}
run 600 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
prof() {  # prof <tag> <bench args...>
  local tag=$1; shift
  run 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/bench.py --cpu-baseline-updates 0 "$@" > $O/${tag}_under_rocprof.json 2> $O/${tag}_rocprof.err
  cp $(ls $O/stats_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $O/stats_$tag
}
prof bench
prof bench_nofused --no-fused-mlp
prof bench_forcedp --force-dp
prof bench_wide_3x128 --hidden-dim 128 --num-layers 3 --steps 30
prof bench_wide_3x64 --hidden-dim 64 --num-layers 3 --steps 30
prof bench_wide_2x128 --hidden-dim 128 --num-layers 2 --steps 30
for shape in "128 3" "64 3" "128 2"; do
  set -- $shape
  run 600 python3 $R/bench.py --hidden-dim $1 --num-layers $2 --steps 30 --cpu-baseline-updates 1 > $O/bench_wide_$2x$1.json 2> $O/bench_wide_$2x$1.err
done
run 600 python3 $R/tools/bench_wide.py > $O/wide_bench.json 2> $O/wide_bench.err
run 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 > /dev/null 2>&1
run 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, json
vals = {}
for d in ("$O/pmc_fetch", "$O/pmc_write"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp_step2" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
M = 131072
alg = M * 300           # observation row 256 B + action row 24 B + record 16 B + index 4 B
res = {"kernel": "k_mlp_step2", "launches": {k: len(v) for k, v in vals.items()},
       "fetch_size_raw_bytes": fetch, "write_size_bytes": write,
       "fetch_corrected_bytes": 2 * fetch, "hbm_bytes_per_launch": 2 * fetch + write,
       "algorithmic_read_bytes": alg,
       "algorithmic_note": "obs 256 B + action 24 B + record 16 B + idx 4 B per sample, read once; writes = gradient slabs (one per workgroup)",
       "correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md section HBM: gfx950 tallies a 128-B memory-side request at 64 B); WRITE_SIZE exact. Every random access moves whole 128-B lines: observation row 2 lines + packed 64-B record (record and action row) 1 line = 384 B per sample against 300 B algorithmic",
       "collected": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --steps 2 --warmup 2 --no-probe; mean over the launches; counters in KiB"}
json.dump(res, open("$O/mlp_pmc.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $O
