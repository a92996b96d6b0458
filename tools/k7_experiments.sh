#!/bin/bash
# What bounds k_mlp_step3?  Builds of the library that do LESS of one kind of work (their results are wrong; only their time is
# of interest), timed against the product build with tools/k7_time.py at M = 131 072:
#   -DK7_EXP_HALF_MFMA    three of the six bf16 products of every 32x32 block (63 of a tile's 126 v_mfma_f32_32x32x16_bf16 per wave)
#   -DK7_EXP_CHEAP_SPLIT  3 of the 11 vector instructions of every three-way split (~290 of a tile's ~1 560 vector instructions per wave)
# Round 4 read (profiles/r04/k7_experiments.txt): 122.7 us -> 103.9 (half the products) and 113.6 (cheap split): the launch moves
# by ~0.85 x (matrix-pipe cycles removed) and ~0.75 x (4 cycles per vector instruction removed), summed over the two waves of a
# SIMD -- matrix and vector work of the two co-resident waves add up rather than overlap (SQ_VALU_MFMA_COEXEC_CYCLES is 18 % of
# the matrix-pipe time, profiles/r04/mlp3_mfma_pmc.json) -- whereas moving work between phases, hiding weight-gradient chains
# under epilogues (coarsely or one instruction at a time), spreading the weight loads or staggering the two tile sets moved nothing.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/build_variant.sh ab_half.so -DK7_EXP_HALF_MFMA || exit 1
tools/build_variant.sh ab_split.so -DK7_EXP_CHEAP_SPLIT || exit 1
for i in 1 2; do
  for v in "" ab_half.so ab_split.so; do
    if [ -n "$v" ]; then export AURPPO_LIB=$R/$v; else unset AURPPO_LIB; fi
    echo "lib=${v:-product} $(timeout -k 10 300 python3 tools/k7_time.py 2>&1 | tail -1)"
  done
done
