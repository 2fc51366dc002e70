#!/bin/bash
# per-kernel times of the given workloads (quick look during kernel work): bash tests/kt_gpu.sh "flipout_conv_s10 radial_conv_s20" [bench flags]
for wl in ${1:-flipout_conv_s10}; do
  python bench.py --workload $wl --steps 50 --warmup 3 --no-cpu-baseline --no-companions ${@:2} 2>/dev/null | python -c "
import json, sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(d['config']['workload'], d['dtype'], 'ms', round(d['ms_per_step'], 4), d['roofline']['kernel'], round(d['roofline']['frac'], 4))
        print('  ', d['kernel_ms_per_step'])
"
done
