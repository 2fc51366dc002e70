"""Diagnostics: timeline of workgroup 0 of tf_fwd_kernel from a -DTF_STAMPS=1 build (gpurun_out/tf_stamps.bin, written by
the library after every launch): per wave, averaged over the steady-state window steps, the s_memtime ticks spent in the
loader part, in each job's MFMA loop and epilogue, and waiting at the step barrier."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tf_stamps.bin", dtype=np.uint64).reshape(8, 48, 16).astype(np.int64)
lo, hi = 6, 34
names = ["load", "j0 mfma", "j0 epi", "j1 mfma", "j1 epi", "j2 mfma", "j2 epi", "barrier"]
step = np.diff(a[:, lo:hi + 1, 0], axis=1).mean()
print(f"ticks per step (wave average) {step:.0f}")
print("wave " + " ".join(f"{n:>8s}" for n in names) + "    busy   step")
for w in range(8):
    rows = []
    for t in range(lo, hi):
        st = a[w, t, :9].copy()
        for ph in range(1, 9):      # a job that does not exist leaves no stamp
            if st[ph] == 0:
                st[ph] = st[ph - 1]
        rows.append(np.diff(st))
    d = np.mean(rows, axis=0)
    nxt = np.mean([a[w, t + 1, 0] - a[w, t, 0] for t in range(lo, hi)])
    print(f"{w:4d} " + " ".join(f"{v:8.0f}" for v in d) + f" {d[:-1].sum():7.0f} {nxt:6.0f}")
# skew: when does each wave pass phase 0 / reach the barrier relative to wave 0
t = 20
print("step", t, "start offsets", (a[:, t, 0] - a[:, t, 0].min()).tolist())
print("step", t, "barrier arrival", (a[:, t, 7] - a[:, t, 0].min()).tolist())
print("step", t, "barrier exit   ", (a[:, t, 8] - a[:, t, 0].min()).tolist())
# k-blocks of job 1 (phases 9..15 = start of k-blocks 0..6)
for w in range(8):
    d = [np.diff(a[w, t, 9:16]) for t in range(lo, hi) if a[w, t, 9] and a[w, t, 15]]
    if d:
        print("wave", w, "j1 k-block ticks", np.mean(d, axis=0).round().tolist())
