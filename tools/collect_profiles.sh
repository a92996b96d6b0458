#!/bin/bash
# Regenerates the round's committed evidence on the GPU box (outputs under gpurun_out/prof_final/; copy into profiles/rNN/):
#   bench.json                          python bench.py (default flags: parity gate + cpu_baseline + both rooflines)
#   bench_under_rocprof.json + bench_kernel_stats.csv            the same workload under rocprofv3 --kernel-trace --stats, with
#                                       --shard-envs-per-gpu 0: the 512-env shard bench.py measures second in the same process launches the
#                                       same kernels at M = 16 384 and would be averaged into the same rows
#   bench_shard_*                       that shard alone (--envs-per-gpu 512), same way; bench_default_*: the default flags (both workloads, mixed rows)
#   bench_nofused_*.json/csv            bench.py --no-fused-mlp (per-op path: K1, K3, K4+K5, K6b stand-alone durations)
#   bench_forcedp_*.json/csv            bench.py --force-dp (the W > 1 launch path captured around a one-rank RCCL all-reduce)
#   mlp3_pmc.json                       separate --pmc passes for the default K7 kernel's (k_mlp_step3) HBM traffic; mlp_pmc.json: k_mlp_step2's
#   k7_stamps_v{3,2}.txt                in-kernel phase stamps of the two K7 builds (tools/mlp_stamps.py); k7_time.txt: stand-alone launch times
#   k2_stamps.txt, k2_time.txt          accept-kernel stamps and the shuffle pipeline's stand-alone time for both accept kernels
#   bench_wide_{3x128,3x64,2x128}*.json/csv   bench.py --hidden-dim/--num-layers (K7w / K8w), plain and under rocprofv3
#   wide_bench.json                     tools/bench_wide.py: K7w / K8w against the per-op path over the -d / -nl shapes
#   (PART=4) k7w_stamps_*.txt, k7w_time.txt, conv_bench_*.json, wgrad_bench_*.json, conv_wgrad_k12_mfma_pmc.json, bench_robot{3,5}.json, equiv5_k11_k12.json
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
# PART=1 | 2 | 3 | 4 runs a part of the list (a gpurun call is capped at 20 minutes); default: everything
part() { [ "${PART:-all}" = all ] || [ "$PART" = "$1" ]; }
export TMPDIR=/tmp
prof() {  # prof <tag> <bench args...>
  local tag=$1; shift
  cd /tmp
  run 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/bench.py --cpu-baseline-updates 0 "$@" > $O/${tag}_under_rocprof.json 2> $O/${tag}_rocprof.err
  cp $(ls $O/stats_$tag/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $O/stats_$tag
  cd $R
}
if part 1; then
run 600 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json; echo
prof bench --shard-envs-per-gpu 0
prof bench_shard --envs-per-gpu 512 --shard-envs-per-gpu 0
prof bench_default
prof bench_nofused --no-fused-mlp
prof bench_forcedp --force-dp
fi
if part 2; then
prof bench_wide_3x128 --hidden-dim 128 --num-layers 3 --steps 30
prof bench_wide_3x64 --hidden-dim 64 --num-layers 3 --steps 30
prof bench_wide_2x128 --hidden-dim 128 --num-layers 2 --steps 30
for shape in "128 3" "64 3" "128 2"; do
  set -- $shape
  run 600 python3 $R/bench.py --hidden-dim $1 --num-layers $2 --steps 30 --cpu-baseline-updates 1 > $O/bench_wide_$2x$1.json 2> $O/bench_wide_$2x$1.err
done
run 600 python3 $R/tools/bench_wide.py > $O/wide_bench.json 2> $O/wide_bench.err
fi
if part 3; then
# (every step below runs under `run`, i.e. its own timeout; a pipe into tail would swallow its exit code, so outputs go to files first)
for v in 3 2; do AURPPO_K7_VARIANT=$v run 300 python3 $R/tools/mlp_stamps.py > $O/k7_stamps_v$v.txt 2>&1; done
: > $O/k7_time.txt
for m in 131072 16384; do for v in 3 2; do K7_M=$m AURPPO_K7_VARIANT=$v run 300 python3 $R/tools/k7_time.py > $O/.step.txt 2>&1; tail -1 $O/.step.txt >> $O/k7_time.txt; done; done
: > $O/k2_time.txt
for a in 3 1; do AURPPO_K2_ACCEPT=$a run 300 python3 $R/tools/k2_time.py > $O/.step.txt 2>&1; tail -1 $O/.step.txt >> $O/k2_time.txt; done
: > $O/k2_stamps.txt
for a in 3 1; do AURPPO_K2_ACCEPT=$a run 300 python3 $R/tools/accept_stamps.py > $O/.step.txt 2>&1; tail -10 $O/.step.txt >> $O/k2_stamps.txt; done
K7_M=16384 AURPPO_K7_VARIANT=3 run 300 python3 $R/tools/mlp_stamps.py > $O/.step.txt 2>&1; tail -20 $O/.step.txt > $O/k7_stamps_v3_M16384.txt
rm -f $O/.step.txt
cd /tmp
pmc() {  # pmc <variant> <kernel> <out>
  rm -rf $O/pmc_fetch $O/pmc_write
  AURPPO_K7_VARIANT=$1 run 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 --no-parity --shard-envs-per-gpu 0 > /dev/null 2>&1
  AURPPO_K7_VARIANT=$1 run 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 --no-parity --shard-envs-per-gpu 0 > /dev/null 2>&1
  python3 $R/tools/parse_pmc.py $2 $O/$3 $O/pmc_fetch $O/pmc_write
  rm -rf $O/pmc_fetch $O/pmc_write
}
pmc 3 k_mlp_step3 mlp3_pmc.json
pmc 2 k_mlp_step2 mlp_pmc.json
# matrix-pipe / wave-state counters (tools/pmc_mfma.sh: its own rocprofv3 passes, --pmc + --kernel-trace only)
cd $R
run 900 bash tools/pmc_mfma.sh k7 k_mlp_step3 72351744 bench.py --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 --no-parity --shard-envs-per-gpu 0 > /dev/null 2>&1
cp gpurun_out/k7_mfma_pmc.json $O/mlp3_mfma_pmc.json
run 900 bash tools/pmc_mfma.sh wide3x128 k_mlpw 0 bench.py --hidden-dim 128 --num-layers 3 --steps 2 --warmup 2 --no-probe --cpu-baseline-updates 0 --no-parity --shard-envs-per-gpu 0 > /dev/null 2>&1
cp gpurun_out/wide3x128_mfma_pmc.json $O/mlp_wide3_3x128_mfma_pmc.json
fi
if part 4; then
# round 4: the wide fused step's stamps / stand-alone times, the convolution kernels' benches and counters, the robot updates
cd $R
for shape in "128 3" "128 2"; do set -- $shape; run 300 python3 $R/tools/k7w_stamps.py $1 $2 > $O/k7w_stamps_$2x$1.txt 2>&1; done
: > $O/k7w_time.txt
for sh in 128,3,64 128,3,128 128,2,64 128,2,128 96,2,64 128,1,64; do K7W_SHAPE=$sh run 120 python3 $R/tools/k7w_time.py > $O/.step.txt 2>&1; tail -1 $O/.step.txt >> $O/k7w_time.txt; done
run 300 python3 $R/tools/bench_conv.py --batch 2048 > $O/conv_bench_b2048.json 2> /dev/null
run 300 python3 $R/tools/bench_wgrad.py --batch 8192 > $O/wgrad_bench_b8192.json 2> /dev/null
run 300 python3 $R/tools/bench_wgrad.py --batch 4096 --size 84 > $O/wgrad_bench_b4096_84.json 2> /dev/null
run 300 python3 $R/tools/bench_wgrad.py --batch 4096 --size 84 --equiv > $O/wgrad_bench_b4096_equiv.json 2> /dev/null
run 200 python3 $R/tools/bench_wgrad.py --linear > $O/wgrad_bench_linear.json 2> /dev/null
run 900 bash tools/pmc_mfma.sh k12 k_conv3x3_wgrad 0 tools/bench_wgrad.py --batch 2048 > /dev/null 2>&1
cp gpurun_out/k12_mfma_pmc.json $O/conv_wgrad_k12_mfma_pmc.json
run 600 python3 $R/bench.py --workload robot3 --steps 5 --warmup 3 > $O/bench_robot3.json 2> $O/bench_robot3.err
run 900 python3 $R/bench.py --workload robot5 > $O/bench_robot5.json 2> $O/bench_robot5.err
run 900 python3 $R/tools/bench_robot.py --config 5 --equivariant --updates 2 --kernel-table > $O/equiv5_k11_k12.json 2> $O/equiv5.err
rm -f $O/.step.txt
fi
cd $R
ls -la $O
