#!/bin/bash
# bench lines of every workload (quick view; the judged line is `python bench.py`)
for wl in lrt_linear_s1 lrt_conv_s1 radial_conv_s20 flipout_conv_s10; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$wl', 'ms', round(d['ms_per_step'],3), 'value', round(d['value']), d['roofline']['kernel'], round(d['roofline']['frac'],4)); print('   ', d['kernel_ms_per_step'])
"
done
