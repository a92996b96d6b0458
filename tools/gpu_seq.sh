#!/bin/bash
# Run GPU steps one after another on the gpurun box: each under its own `timeout -k 10`, logging to gpurun_out/;
# a step that fails with an ordinary error does not stop the sequence, a step that TIMED OUT or was KILLED does
# (no further GPU step after a hang).   usage: tools/gpu_seq.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/$name.out"; tail -n 4 "gpurun_out/$name.err"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out / was killed: stopping"; exit $rc; fi
done
exit 0
