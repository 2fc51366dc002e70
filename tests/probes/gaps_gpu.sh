#!/bin/bash
# Diagnostics: are there launch gaps between the kernels of a step?  rocprofv3 kernel trace of a bench run; prints, for the timed
# steps, the sum of kernel durations per step and the wall time per step (first kernel start to next step's first start).
#   bash tests/probes/gaps_gpu.sh lrt_conv_s1 --batch 100
WL=${1:-flipout_conv_s10}; shift
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/gaps
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline --no-companions "$@" > gpurun_out/gaps.log 2>&1
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/gaps/**/*_kernel_trace.csv', recursive=True))[-1]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# a step starts at its first kernel: the noise / inputs / sampling launch
first = [i for i, r in enumerate(rows) if 'step_inputs' in r[2] or 'gen_' in r[2] or 'x_planes' in r[2] or 'xf_planes' in r[2]]
if not first:
    first = [i for i, r in enumerate(rows) if 'prep_' in r[2]]
starts = [first[0]] + [i for a, i in zip(first, first[1:]) if i - a > 3]
steps = []
for a, b in zip(starts[10:50], starts[11:51]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    wall = rows[b][0] - seg[0][0]
    steps.append((busy, wall, len(seg)))
import statistics
print('kernels per step', statistics.median(s[2] for s in steps), 'busy us', statistics.median(s[0] for s in steps) / 1e3,
      'wall us', statistics.median(s[1] for s in steps) / 1e3)
seg = rows[starts[20]:starts[21]]
t0 = seg[0][0]
for (s, e, n), nxt in zip(seg, seg[1:] + [rows[starts[21]]]):
    print(f'{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  gap after {(nxt[0] - e) / 1e3:6.1f}  {n[:70]}')
PY
