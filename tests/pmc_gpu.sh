#!/bin/bash
mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1; echo EXIT $? >> gpurun_out/pmc_sq.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq2.log 2>&1; echo EXIT $? >> gpurun_out/pmc_sq2.log
tail -2 gpurun_out/pmc_sq.log gpurun_out/pmc_sq2.log | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc_sq','pmc_sq2'):
    fs = glob.glob(f'gpurun_out/{d}/*/*_counter_collection.csv')
    if not fs: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        if 'conv_' in k or 'dense_' in k or 'pool' in k or 'trunk' in k:
            print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
