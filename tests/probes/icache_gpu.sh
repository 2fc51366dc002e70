#!/bin/bash
# Diagnostics: instruction-cache counters of the step's kernels (is the unrolled, role-specialised code thrashing it?)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > gpurun_out/counters_avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*" gpurun_out/counters_avail.txt | sort -u | tr '\n' ' '; echo
rm -rf gpurun_out/pmc_ic
ARGS="--workload ${1:-flipout_conv_s10} --prec f32 --no-cpu-baseline --no-companions --steps 4 --warmup 1"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_ic -- python3 bench.py $ARGS > gpurun_out/pmc_ic.log 2>&1; echo EXIT $? >> gpurun_out/pmc_ic.log
tail -2 gpurun_out/pmc_ic.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_ic/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))[:14]:
    n = max(cnt[k], 1)
    print(k, {c: round(x / n) for c, x in v.items()})
PY
