#!/usr/bin/env python3
"""Assembly kernel times (HIP events) for the three FEM batches of bench.py.  usage: fem_asm_prof.py [nmesh]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from orb_slam2_e_amd.fem import FEA2, FEA2Batch, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch, synth_tet_batch_distinct
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for label in ("single", "batch", "distinct"):
    if label == "distinct":
        n, t, f, l = synth_tet_batch_distinct(nm)
        fea = FEA2Batch(n, t, FEM_TET4)
    else:
        n, t, f, l = synth_tet_batch(1 if label == "single" else nm, 12)
        fea = FEA2(n, t, FEM_TET4)
    fea.MatrixAssembly()
    fea.profile(True)
    t0 = time.perf_counter()
    for _ in range(5): fea.MatrixAssembly()
    dt = (time.perf_counter() - t0) / 5
    print(label, "wall ms %.3f" % (dt * 1e3), {k: round(v[0] / max(v[1], 1), 4) for k, v in fea.profile_read().items() if v[1]})
