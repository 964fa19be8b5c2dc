#!/bin/bash
# A/B timing of library builds on ONE box: tools/gpu_ab.sh libA.so libB.so ... (alternating, 3 rounds).
# A build that fails to load or to run shows its stderr and a FAILED line instead of a JSON traceback.
for round in 1 2 3; do
  for lib in "$@"; do
    out=$(CLIMA_HIP_LIB=$PWD/$lib python3 bench.py --steps 400 --warmup 200 --repeats 3 --no-cpu-baseline $AB_ARGS)
    rc=$?
    if [ $rc -ne 0 ] || [ -z "$out" ]; then echo "$lib round $round FAILED (exit $rc)"; continue; fi
    echo "$out" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'round $round', '%.2f us/call' % (1e3*d['ms_per_step']), d['roofline']['kernel_us'], 'sync %.1f' % d['sync_api']['us_median'] if d.get('sync_api') else '')"
  done
done
