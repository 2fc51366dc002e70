#!/bin/bash
# Diagnostics: timeline of tf_fwd_kernel's workgroup 0 (s_memtime stamps) for the variants given as extra -D flags, e.g.
#   bash tests/probes/stamps_gpu.sh "" "-DTFX=1"        (ABL_WL selects the workload)
mkdir -p gpurun_out
i=0
for extra in "$@"; do
  ABL_CFLAGS="-Xclang -target-feature -Xclang -packed-fp32-ops -DTF_STAMPS=1 $extra" bash tests/probes/ablate_gpu.sh "" 2>&1 | grep "^base"
  echo "== stamps [$extra] ${ABL_WL:-flipout_conv_s10}"
  python tests/probes/tf_stamps.py
  cp gpurun_out/tf_stamps.bin gpurun_out/tf_stamps_$i.bin; i=$((i+1))
done
