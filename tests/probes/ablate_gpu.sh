#!/bin/bash
# Diagnostics (not part of the product): builds variants of the library with parts of the fused trunk kernels removed
# (-DTR_ABL / -DTX_ABL / -DDK_ABL bit masks, results wrong; -DTR_PERM=w0,...,w11 places the trunk forward roles on other
# waves, results right) and times the step with each.  Run on the GPU box:
#   bash tests/probes/ablate_gpu.sh "TR_ABL=1 TR_ABL=2 TX_ABL=1 ..."
mkdir -p gpurun_out /tmp/abl
cd bayesrul_amd/csrc
for v in base $1; do
  flag=""; [ "$v" != base ] && flag="-D$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC ${ABL_CFLAGS--Xclang -target-feature -Xclang -packed-fp32-ops} $flag -o /tmp/abl/lib_$v.so plan.hip &
done
wait
cd ../..
for v in base $1; do
  python - <<PY
import json, subprocess, sys, os
sys.path.insert(0, '.')
import bayesrul_amd._native as N
N.LIB_PATH = '/tmp/abl/lib_$v.so'
sys.argv = ['bench.py', '--steps', '50', '--warmup', '3', '--no-cpu-baseline', '--no-companions', '--workload', os.environ.get('ABL_WL', 'flipout_conv_s10')]
import io, contextlib, runpy
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path('bench.py', run_name='__main__')
for line in buf.getvalue().splitlines():
    if line.startswith('{'):
        d = json.loads(line)
        k = d['kernel_ms_per_step']
        if os.environ.get('ABL_WL'): print('$v', 'ms', round(d['ms_per_step'], 4), k)
        print('$v', 'ms', round(d['ms_per_step'], 3), 'fwd0', k.get('fwd[0]'), 'dx1', k.get('dx[1]'), 'dw', k.get('dw[0]'), k.get('dw[1]'), k.get('dw[2]'), 'fwd3', k.get('fwd[3]'), 'dx3', k.get('dx[3]'))
PY
done
