#!/bin/bash
# round-end evidence run: the judged bench line (with companions + cpu_baseline), one bench line per workload and precision
# plan, and the rocprofv3 passes of the main workloads.  Raw output under gpurun_out/; afterwards
# `python profiles/make_summary.py r03_<workload>_<prec> gpurun_out/keep_<workload>_<prec>` per set.
# (a gpurun call is limited to 20 minutes: `SKIP_BENCH=1 bash tests/round_gpu.sh "wl:prec wl:prec"` runs profile jobs only)
mkdir -p gpurun_out
if [ -z "$SKIP_BENCH" ]; then
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
bash tests/wl_gpu.sh > gpurun_out/wl_all.log 2>&1; cat gpurun_out/wl_all.log
for wl in flipout_conv_s10 radial_conv_s20 predict_conv_s100; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-companions --workload $wl --prec bf16x3 2>/dev/null > gpurun_out/wl_${wl}_bf16x3.json
done
fi
[ -n "$SKIP_PROF" ] && exit 0
for job in ${1:-flipout_conv_s10:f32 flipout_conv_s10:bf16x3 radial_conv_s20:f32 predict_conv_s100:f32 lrt_conv_s1:f32 lrt_linear_s1:bf16x3}; do
  wl=${job%%:*}; prec=${job##*:}
  bash tests/prof_gpu.sh $wl $prec > gpurun_out/prof_${wl}_${prec}.log 2>&1
  mkdir -p gpurun_out/keep_${wl}_${prec}
  for d in prof_stats prof_fetch prof_write pmc_sq pmc_sq2; do
    mkdir -p gpurun_out/keep_${wl}_${prec}/$d
    find gpurun_out/$d -name "*.csv" -size -20000k -exec cp {} gpurun_out/keep_${wl}_${prec}/$d/ \;
    cp gpurun_out/$d.log gpurun_out/keep_${wl}_${prec}/ 2>/dev/null
  done
  echo "profiled $wl $prec"
done
