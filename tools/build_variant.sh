#!/bin/bash
# tools/build_variant.sh <out.so> [-DFLAG ...]: the library with extra defines, for timing experiments (AURPPO_LIB=<out.so> tools/k7_time.py)
out=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/aur_ppo_amd/csrc
srcs=$(python3 -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; print(' '.join(g.HIP_SOURCES))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off "$@" $srcs -o $R/$out
