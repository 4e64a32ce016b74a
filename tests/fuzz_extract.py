"""Differential fuzz: HIP extractor vs CPU oracle on random image sizes / parameters / contents.
usage: fuzz_extract.py [ncases] [seed]"""
import sys
import numpy as np
import oracle
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
skipped = 0
why = {}
for case in range(n):
    w = int(rng.integers(220, 900)); h = int(rng.integers(180, 700))
    if rng.random() < 0.25:      # wide and tall frames: several initial octree nodes per level (or the reference's division by zero)
        w, h = (int(rng.integers(500, 1400)), int(rng.integers(110, 300))) if rng.random() < 0.7 else (int(rng.integers(150, 300)), int(rng.integers(300, 800)))
    nlev = int(rng.integers(1, 9)); sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0]))
    if rng.random() < 0.15:      # unusual pyramids: nearly equal levels, very steep ones, more than eight levels
        nlev = int(rng.integers(1, 17)); sf = float(rng.choice([1.03, 1.07, 1.15, 1.4, 1.7, 2.5, 3.0]))
    nfeat = int(rng.integers(100, 3500)) if rng.random() < 0.7 else int(rng.integers(1, 100))    # tiny quotas: 0 .. 3 per level
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(3, ini + 1))
    kind = rng.integers(0, 4)
    if kind == 0: img = synth_frame(int(rng.integers(0, 10000)), w, h)
    elif kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 2:
        img = synth_frame(int(rng.integers(0, 10000)), w, h); img = (img // 4 + 100).astype(np.uint8)          # low contrast
    else:
        img = np.full((h, w), 90, np.uint8)
        for _ in range(int(rng.integers(1, 60))):
            x, y = rng.integers(0, w - 20), rng.integers(0, h - 20)
            img[y:y + rng.integers(3, 20), x:x + rng.integers(3, 20)] = rng.integers(0, 256)
    params = (nfeat, sf, nlev, ini, mn)
    try:
        o = oracle.OrbOracle(*params)
    except Exception as e:
        print("oracle rejects", params, w, h, e); continue
    view = img
    if rng.random() < 0.3:       # the image as a region of interest of a wider, taller array: a row pitch (aligned or not), poisoned surroundings
        padx, pady, offx, offy = int(rng.integers(1, 70)), int(rng.integers(0, 5)), int(rng.integers(0, 9)), int(rng.integers(0, 3))
        big = np.full((h + pady + offy, w + padx + offx), 0x5A, np.uint8)
        big[offy:offy + h, offx:offx + w] = img
        view = big[offy:offy + h, offx:offx + w]
    try:
        ex = ORBextractor(*params)
        kps, desc = ex(view)
    except Exception as e:
        # geometry the reference cannot run either (cell grid / nIni = 0) must be rejected by both
        if "error -5" in str(e):      # documented capacity limit (per-level quota above 2047, cell larger than the LDS tile)
            skipped += 1; why[str(e).split(":", 1)[-1].strip()] = why.get(str(e).split(":", 1)[-1].strip(), 0) + 1; continue
        try:
            o.extract(img); print("MISMATCH: gpu rejects, oracle runs", params, w, h, e); bad += 1
        except Exception:
            pass
        continue
    okps, odesc = o.extract(img)
    ok = len(kps) == len(okps) and np.array_equal(desc, odesc) and all(
        np.array_equal(kps[f].view(np.uint32) if kps[f].dtype.kind == "f" else kps[f],
                       okps[f].view(np.uint32) if okps[f].dtype.kind == "f" else okps[f]) for f in kps.dtype.names)
    if not ok:
        bad += 1
        print("MISMATCH", case, params, (w, h), "kind", kind, len(kps), len(okps), flush=True)
    if rng.random() < 0.25:      # the batch entry on the same handle: B frames of this size (B % 8 == 0 takes the XCD-aware mapping)
        B = int(rng.choice([1, 2, 3, 5, 6, 7, 8, 9, 16, 24]))   # (up to 6: the small-batch kernel forms, k_pyr_chain and the 512-thread k_octree)
        imgs = np.stack([img] + [np.roll(img, int(rng.integers(1, 40)) * (b + 1), axis=int(rng.integers(0, 2))) for b in range(B - 1)])
        ex.extract_batch(imgs)
        kb, db, cb = ex.download_batch()
        for b in range(B):
            rk, rd = (okps, odesc) if b == 0 else o.extract(imgs[b])
            nb_ = int(cb[b])
            if not (nb_ == len(rk) and np.array_equal(db[b, :nb_], rd) and kb[b, :nb_].tobytes() == rk.tobytes()):
                bad += 1
                print("MISMATCH batch", case, params, (w, h), "B", B, "frame", b, nb_, len(rk), flush=True)
                break
print("unsupported:", why)
print("cases", n, "skipped (unsupported sizes)", skipped, "mismatches", bad)
sys.exit(1 if bad else 0)
