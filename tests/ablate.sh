#!/bin/bash
for a in 0 1 2 4 8 15; do
  BNN_DW_ABLATE=$a timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['kernel_ms_per_step']; print('dw ablate $a', {x:k[x] for x in k if x.startswith('dw')})
"
done
