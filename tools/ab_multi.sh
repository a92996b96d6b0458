#!/bin/bash
# A/B several builds of the library on one box: tools/ab_multi.sh rounds a.so b.so ...; round-robin so drift cancels.
# Leaves the library that was in place when it started.
rounds=$1; shift
cp aur_ppo_amd/libaurppo_hip.so /tmp/ab_keep.so
for r in $(seq $rounds); do
  for so in "$@"; do
    cp $so aur_ppo_amd/libaurppo_hip.so
    line=$(timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-baseline-updates 0 2>/dev/null | tail -1)
    python - "$so" "$line" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); r = d["roofline"]
    print(f"{sys.argv[1]:14s} ms_per_step {d['ms_per_step']:.4f}  probe_us {r.get('avg_launch_us', 0)}  frac {r['frac']:.4f}  parity {d.get('parity_checked')}")
except Exception as e:
    print(f"{sys.argv[1]:14s} FAILED ({sys.argv[2][:80]!r})")
PY
  done
done
cp /tmp/ab_keep.so aur_ppo_amd/libaurppo_hip.so
