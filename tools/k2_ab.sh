#!/bin/bash
# tools/k2_ab.sh lib.so : shuffle parity tests + stand-alone pipe timing + in-situ side-stream period with that build
cp $1 aur_ppo_amd/libaurppo_hip.so
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "shuffle or perm or state" --timeout 300 2>&1 | tail -1
python tools/bench_shuffle_pipe.py
python bench.py --steps 40 --warmup 5 --cpu-baseline-updates 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],4), d['roofline']['side_stream']['k2_period_ms'], d['roofline']['side_stream']['slack_ms'])"
