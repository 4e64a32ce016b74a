"""Overlap analysis of a rocprofv3 --kernel-trace CSV: for the steady part of a pipelined bench run, how long each kernel
kind is on the GPU (union of its dispatch intervals), how much of that time it runs alone, and the mean number of
kernels in flight.  usage: timeline.py <kernel_trace.csv> [reference kernel]"""
import csv, sys, re
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]; m = re.search(r"(k_[a-z0-9_]+)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else n[:24]))
rows.sort()
# steady window: between 30 % and 80 % of the dispatches of the kernel named in argv[2] (default k_fast_cells)
ref = sys.argv[2] if len(sys.argv) > 2 else "k_fast_cells"
starts = [s for s, e, k in rows if k == ref]
lo, hi = starts[int(len(starts) * 0.3)], starts[int(len(starts) * 0.8)]
nsteps = int(len(starts) * 0.8) - int(len(starts) * 0.3)
rows = [r for r in rows if lo <= r[0] < hi]
ev = []
for s, e, k in rows:
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
active = defaultdict(int); nact = 0; last = ev[0][0]
busy = 0; conc_time = 0; alone = defaultdict(int); on = defaultdict(int)
for t, d, k in ev:
    dt = t - last
    if nact > 0:
        busy += dt; conc_time += dt * nact
        kinds = [q for q, c in active.items() if c > 0]
        for q in kinds:
            on[q] += dt
            if len(kinds) == 1: alone[q] += dt
    active[k] += d; nact += d; last = t
span = ev[-1][0] - ev[0][0]
print(f"window {span/1e6:.2f} ms = {nsteps} steps ({span/1e6/nsteps:.4f} ms/step), GPU busy {100*busy/span:.1f} %, mean kernels in flight while busy {conc_time/max(busy,1):.2f}")
print(f"{'kernel':22s} {'dispatches':>10s} {'on GPU %':>9s} {'alone %':>8s} {'mean us':>8s}")
cnt = defaultdict(int); dur = defaultdict(int)
for s, e, k in rows: cnt[k] += 1; dur[k] += e - s
for k in sorted(on, key=lambda q: -on[q]):
    print(f"{k:22s} {cnt[k]:10d} {100*on[k]/span:9.1f} {100*alone[k]/span:8.1f} {dur[k]/cnt[k]/1e3:8.1f}")
