#!/bin/bash
# bench + rocprofv3 evidence of one workload (default flipout_conv_s10): kernel stats, HBM traffic (separate FETCH_SIZE /
# WRITE_SIZE passes), SQ counters (two passes).  Raw output under gpurun_out/ (scratch); `python profiles/make_summary.py
# <tag>` turns it into the committed files under profiles/.
WL=${1:-flipout_conv_s10}
PREC=${2:-f32}   # pass bf16x3 for the split-bf16 plan (lrt_linear_s1 runs on it)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_sq gpurun_out/pmc_sq2
ARGS="--workload $WL --prec $PREC --no-cpu-baseline --no-companions"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 20 --warmup 3 $ARGS > gpurun_out/prof_stats.log 2>&1; echo EXIT $? >> gpurun_out/prof_stats.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 bench.py --steps 5 --warmup 1 $ARGS > gpurun_out/prof_fetch.log 2>&1; echo EXIT $? >> gpurun_out/prof_fetch.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 bench.py --steps 5 --warmup 1 $ARGS > gpurun_out/prof_write.log 2>&1; echo EXIT $? >> gpurun_out/prof_write.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 4 --warmup 1 $ARGS > gpurun_out/pmc_sq.log 2>&1; echo EXIT $? >> gpurun_out/pmc_sq.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 4 --warmup 1 $ARGS > gpurun_out/pmc_sq2.log 2>&1; echo EXIT $? >> gpurun_out/pmc_sq2.log
tail -q -n 1 gpurun_out/prof_stats.log gpurun_out/prof_fetch.log gpurun_out/prof_write.log gpurun_out/pmc_sq.log gpurun_out/pmc_sq2.log
python3 profiles/make_summary.py scratch_$WL > /dev/null 2>&1
head -30 profiles/scratch_${WL}_kernel_stats.csv | cut -c1-160
