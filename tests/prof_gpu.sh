#!/bin/bash
# bench + rocprofv3 kernel stats + PMC traffic of the default workload (round-end evidence)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 300 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_final.log 2>&1; echo EXIT $? >> gpurun_out/bench_final.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1; echo EXIT $? >> gpurun_out/prof_stats.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_fetch.log 2>&1; echo EXIT $? >> gpurun_out/prof_fetch.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/prof_write.log 2>&1; echo EXIT $? >> gpurun_out/prof_write.log
grep -h "^{" gpurun_out/bench_final.log | cut -c1-1800
tail -2 gpurun_out/prof_fetch.log gpurun_out/prof_write.log | cut -c1-300
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name "*.csv" | head -20
