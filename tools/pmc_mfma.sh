#!/bin/bash
# usage: tools/pmc_mfma.sh <tag> <kernel-name-substring> <expected MFMA pipe cycles per launch, or 0> <python script + args>
# Matrix-pipe counters of one kernel (BASELINE config 5 asks for "rocprof HBM/MFMA counters"): ONE rocprofv3 pass with
#   --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  (+ --kernel-trace for the durations of the same dispatches)
# and a second one with the wave-state counters (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
# SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES).  Never combined with the hip/hsa/memory-copy trace domains.
# Prints and stores gpurun_out/<tag>_mfma_pmc.json (tools/parse_mfma_pmc.py says how each derived number is formed).
tag=$1; kern=$2; expect=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcm_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python3 $R/"$@" > $O/a.out 2> $O/a.err || { echo "pass A failed (rc $?)"; tail -5 $O/a.err; exit 1; }
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/b -- python3 $R/"$@" > $O/b.out 2> $O/b.err || { echo "pass B failed (rc $?)"; tail -5 $O/b.err; }
cd $R
python3 $R/tools/parse_mfma_pmc.py "$kern" "$expect" $R/gpurun_out/${tag}_mfma_pmc.json $O/a $O/b
rm -rf $O/a $O/b
