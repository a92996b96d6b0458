#!/bin/bash
# A/B two builds of the library on the same box: tools/ab.sh old.so new.so [rounds]; alternates them so drift cancels.
set -e
old=$1; new=$2; rounds=${3:-3}
for r in $(seq $rounds); do
  for which in old new; do
    cp ${!which} aur_ppo_amd/libaurppo_hip.so
    line=$(timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-baseline-updates 0 2>/dev/null | tail -1)
    python - "$which" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print(f"{sys.argv[1]:4s} ms_per_step {d['ms_per_step']:.4f}  probe_us {r.get('avg_launch_us', 0)}  frac {r['frac']:.4f}")
PY
  done
done
cp $new aur_ppo_amd/libaurppo_hip.so
