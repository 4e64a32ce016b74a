#!/usr/bin/env python3
"""Convert a few of the reference's committed surface meshes
(output/PointClouds/pcr_t_f*.vtk: POINTS n float + POLYGONS m 4m) into small
input fixtures for the C3D6 FEM parity tests: four picked by size (fem_mesh_*.npz) and all 853 in one file (fem_meshes_all.npz).  These are data files of the
reference (inputs only -- nothing in the reference tree records results for them,
SURVEY 8c).  Run in the dev container; output: tests/golden/fem_mesh_*.npz."""
import glob
import os
import sys

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/output/PointClouds"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def read_vtk(path):
    tok = open(path).read().split()
    i = tok.index("POINTS"); n = int(tok[i + 1])
    pts = np.array(tok[i + 3:i + 3 + 3 * n], dtype=np.float32).reshape(n, 3)
    j = tok.index("POLYGONS"); m = int(tok[j + 1])
    poly = np.array(tok[j + 3:j + 3 + 4 * m], dtype=np.int64).reshape(m, 4)
    assert (poly[:, 0] == 3).all()
    return pts, poly[:, 1:].astype(np.int32)


files = sorted(glob.glob(os.path.join(SRC, "pcr_t_f*.vtk")))
sizes = []
for f in files:
    with open(f) as fh:
        for line in fh:
            if line.startswith("POINTS"):
                sizes.append(int(line.split()[1])); break
sizes = np.array(sizes)
order = np.argsort(sizes, kind="stable")
targets = {"min": order[0], "median": order[len(order) // 2], "p90": order[int(len(order) * 0.9)],
           "large": order[np.searchsorted(sizes[order], 400)]}
os.makedirs(OUT, exist_ok=True)
for name, idx in targets.items():
    pts, tris = read_vtk(files[idx])
    np.savez_compressed(os.path.join(OUT, f"fem_mesh_{name}.npz"), points=pts, triangles=tris,
                        source=os.path.basename(files[idx]))
    print(name, os.path.basename(files[idx]), pts.shape, tris.shape)

# ... and ALL of them in one file (round 5): concatenated points / triangles with offsets, the frame number of each file name
pts_l, tri_l, frame = [], [], []
for f in files:
    p_, t_ = read_vtk(f)
    pts_l.append(p_); tri_l.append(t_)
    frame.append(int(os.path.basename(f)[len("pcr_t_f"):-len(".vtk")]))
pt_off = np.concatenate([[0], np.cumsum([len(p_) for p_ in pts_l])]).astype(np.int32)
tri_off = np.concatenate([[0], np.cumsum([len(t_) for t_ in tri_l])]).astype(np.int32)
np.savez_compressed(os.path.join(OUT, "fem_meshes_all.npz"), points=np.concatenate(pts_l), triangles=np.concatenate(tri_l),
                    pt_off=pt_off, tri_off=tri_off, frame=np.array(frame, np.int32))
print("all:", len(files), "meshes,", pt_off[-1], "points,", tri_off[-1], "triangles")
