#!/bin/bash
# tools/env_ab.sh "VAR=value" [rounds]: bench.py alternately without and with one environment setting (same library, same box)
setting=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for which in base with; do
    if [ $which = with ]; then line=$(env $setting timeout -k 10 300 python bench.py --steps 60 --warmup 5 --cpu-baseline-updates 0 2>/dev/null | tail -1)
    else line=$(timeout -k 10 300 python bench.py --steps 60 --warmup 5 --cpu-baseline-updates 0 2>/dev/null | tail -1); fi
    python - "$which" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]; ss = r["side_stream"]
print(f"{sys.argv[1]:4s} ms_per_step {d['ms_per_step']:.4f}  K7 {r['avg_launch_us']}  k2_period {ss['k2_period_ms']}  slack {ss['slack_ms']}")
PY
  done
done
