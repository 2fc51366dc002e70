#!/bin/bash
# quick GPU loop used during kernel work: bf16x3 parity tests + bench line
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -k "bf16x3 or full_size or dp_two" > gpurun_out/quick_pytest.log 2>&1
echo EXIT $? >> gpurun_out/quick_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/quick_bench.log 2>&1
echo EXIT $? >> gpurun_out/quick_bench.log
tail -15 gpurun_out/quick_pytest.log
python - <<'PY'
import json
for line in open('gpurun_out/quick_bench.log'):
    if line.startswith('{'):
        d = json.loads(line)
        print('ms/step', round(d['ms_per_step'], 3), 'value', round(d['value']), 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'], 4))
        print(d['kernel_ms_per_step'])
    elif 'EXIT' in line or 'Error' in line or 'error' in line:
        print(line.strip())
PY
