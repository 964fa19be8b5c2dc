#!/bin/bash
# A/B timing of library builds on ONE box: tools/gpu_ab.sh libA.so libB.so ... (alternating, 3 rounds)
for round in 1 2 3; do
  for lib in "$@"; do
    CLIMA_HIP_LIB=$PWD/$lib python bench.py --steps 400 --warmup 200 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'round $round', '%.1f us/call' % (1e3*d['ms_per_step']), d['roofline']['kernel_us'])"
  done
done
