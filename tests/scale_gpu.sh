#!/bin/bash
# per-kernel times vs windows per GPU (slope = per-window cost, intercept = fixed cost per launch)
mkdir -p gpurun_out
for b in 500 1000 2000 4000; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-companions --batch $b 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('B=$b ms', round(d['ms_per_step'],3), d['kernel_ms_per_step'])
"
done
