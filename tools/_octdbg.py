import ctypes as C, numpy as np
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_frame
ex = ORBextractor(2000, 1.2, 8, 20, 7)
img = synth_frame(0)
for _ in range(3): ex(img)
out = np.zeros(64, np.int64)
ex._L.orbx_dbg_oct(out.ctypes.data_as(C.c_void_p))
t0 = out[0]; nr = int(out[60])
print("M", out[61], "S", out[62], "rounds", nr, "total us", (out[59] - t0) / 100.0)
print("setup us", (out[1] - t0) / 100.0)
for r in range(nr):
    b = 2 + 5 * r
    print("round", r, "counts %.1f  pernode %.1f  scan %.1f  mode %.1f  create+relabel %.1f" % tuple(
        (out[b + k + 1 if k < 4 else b + 4] - out[b + k]) / 100.0 if k < 4 else 0 for k in range(5)))
print("tail us", (out[59] - out[58]) / 100.0)
