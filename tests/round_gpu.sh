#!/bin/bash
# round-end evidence run: the judged bench line (with companions + cpu_baseline), one bench line per workload, and the
# rocprofv3 passes of every workload.  Raw output under gpurun_out/; `python profiles/make_summary.py <tag>` afterwards.
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
bash tests/wl_gpu.sh > gpurun_out/wl_all.log 2>&1; cat gpurun_out/wl_all.log
for wl in ${1:-flipout_conv_s10 radial_conv_s20 predict_conv_s100 lrt_linear_s1}; do
  bash tests/prof_gpu.sh $wl > gpurun_out/prof_$wl.log 2>&1
  mkdir -p gpurun_out/keep_$wl
  for d in prof_stats prof_fetch prof_write pmc_sq pmc_sq2; do
    mkdir -p gpurun_out/keep_$wl/$d
    find gpurun_out/$d -name "*.csv" -size -20000k -exec cp {} gpurun_out/keep_$wl/$d/ \;
    cp gpurun_out/$d.log gpurun_out/keep_$wl/ 2>/dev/null
  done
  echo "profiled $wl"
done
