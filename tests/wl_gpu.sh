#!/bin/bash
# bench lines of every workload (quick view; the judged line is `python bench.py`)
mkdir -p gpurun_out
for wl in lrt_linear_s1 lrt_conv_s1 radial_conv_s20 predict_conv_s100 flipout_conv_s10; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-companions --workload $wl ${WL_PREC:+--prec $WL_PREC} 2>/dev/null > gpurun_out/wl_$wl.json
  python - <<PY
import json
for line in open('gpurun_out/wl_$wl.json'):
    if line.startswith('{'):
        d=json.loads(line); print('$wl', 'ms', round(d['ms_per_step'],3), 'value', round(d['value']), d['roofline']['kernel'], round(d['roofline']['frac'],4)); print('   ', d['kernel_ms_per_step'])
PY
done
