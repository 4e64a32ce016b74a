"""Development aid: shader-clock time per phase of k_fast_cells / k_describe / k_pyr_resize per workgroup, and how many
workgroups are resident over a launch (100-MHz wall clock).  Needs a library built with the markers switched on -- never the
product build:
    cd orb_slam2_e_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
        -mllvm -amdgpu-mfma-vgpr-form=1 -DORBX_PHASE_TIMING -c orbx_extract.hip -o /tmp/x.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_ph.so /tmp/x.o orbx_stereo.o orbm_match.o orbm_search.o fem.o
On the GPU box: cp orb_slam2_e_amd/lib_ph.so orb_slam2_e_amd/liborbslam_hip.so && PYTHONPATH=. python3 tools/phase_timing.py
(the markers cost a few per cent and a register or two: compare kernels with tools/trace_ab.sh, not with this build)."""
import ctypes as C, sys
import numpy as np
from orb_slam2_e_amd.extractor import ORBextractor
from orb_slam2_e_amd.synth import synth_sequence

ex = ORBextractor(2000, 1.2, 8, 20, 7)
L = ex._L
frames = synth_sequence(64)
for _ in range(3):
    ex.extract_batch(frames)
ex.download_batch()
acc = np.zeros((4, 65536, 16), np.uint64)
L.orbx_debug_phases.argtypes = [C.c_void_p, C.c_int]
assert L.orbx_debug_phases(acc.ctypes.data, 1) == 0
ex.extract_batch(frames)
ex.download_batch()
assert L.orbx_debug_phases(acc.ctypes.data, 0) == 0
for k, (title, names) in enumerate([("k_fast_cells, per wave", ["tile load", "stage A", "stage B", "zero + score", "nms + emit", "epilogue", "(timer)"]),
                                    ("k_describe, per workgroup (wave 0)", ["lookup", "moments", "atan/sincos", "brief kp0", "brief kp1", "brief kp2", "brief kp3", "tail"]),
                                    ("k_pyr_resize, per wave (records of the launches that wrote last; [12] = level width)", ["row table", "columns + rows + math", "store"]),
                                    ("k_octree, per workgroup = (level, frame); [12] = level, [11] = candidates, [7] = rounds", ["cells + gather", "roots", "children counts", "node flags + scans", "who splits", "new list + keys", "best per leaf"])]):
    full = acc[k]
    used = full[:, :8].sum(axis=1) > 0
    full = full[used]
    if k == 3:
        for lvl in np.unique(full[:, 12]):
            g = full[full[:, 12] == lvl]
            print('   level', int(lvl), 'workgroups', len(g), 'candidates %.0f' % g[:, 11].astype(float).mean(), 'rounds %.1f' % g[:, 7].astype(float).mean(), 'mean clk per phase', g[:, :7].astype(float).mean(axis=0).round().astype(int).tolist(), 'residence %.1f us' % ((g[:, 13] - g[:, 15]).astype(float).mean() / 100))
    if k == 2:
        for wv in np.unique(full[:, 12]):
            g = full[full[:, 12] == wv]
            o = np.argsort(g[:, 15]); q = len(g) // 4
            print('      first-phase clk by start-time quartile:', [int(g[o[i * q:(i + 1) * q], 0].astype(float).mean()) for i in range(4)], 'start spread us', (g[:, 15].max() - g[:, 15].min()) / 100.0)
            for fz in (1, 0):
                gg = g[g[:, 11] == fz]
                if len(gg): print('      %s waves: %d records, mean clk %s, mean residence %.2f us, last end - first start %.2f us' % ('full' if fz else 'tail', len(gg), gg[:, :3].astype(float).mean(axis=0).round(), (gg[:, 13] - gg[:, 15]).astype(float).mean() / 100, (gg[:, 13].max() - g[:, 15].min()) / 100.0))
            print('   level width', int(wv), 'records', len(g), 'mean clk', g[:, :3].astype(float).mean(axis=0).round(), 'span us', (g[:, 13].max() - g[:, 15].min()) / 100.0, 'mean residence us', (g[:, 13] - g[:, 15]).astype(float).mean() / 100)
    r = full[:, :8].astype(np.float64)
    tot = r.sum(axis=1)
    print(title, "records", len(r), "mean total clk", round(tot.mean()), "median", round(np.median(tot)), "p90", round(np.percentile(tot, 90)))
    for i, n in enumerate(names):
        print("   %-14s mean %8.0f clk  %5.1f %%   median %8.0f" % (n, r[:, i].mean(), 100 * r[:, i].sum() / tot.sum(), np.median(r[:, i])))
    # occupancy over the kernel's span (100 MHz wall clock, common to the whole device)
    start = full[:, 15].astype(np.float64); end = full[:, 13].astype(np.float64)
    s0, e0 = start.min(), end.max()
    t = np.linspace(s0, e0, 21)
    print("   span %.1f us, mean resident workgroups %.0f (%.2f per CU), at 5 %% steps: %s" % (
        (e0 - s0) / 100, (end - start).sum() / (e0 - s0), (end - start).sum() / (e0 - s0) / 256, [int(((start <= x) & (end > x)).sum()) for x in t[1:-1]]))
    print("   mean workgroup residence %.2f us" % ((end - start).mean() / 100))
