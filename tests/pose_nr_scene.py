"""Synthetic relocalisation scene for the closed-loop F12 tests: the graph Optimizer::PoseOptimizationNR builds
(src/Optimizer.cc:484-707) around one of the reference's own surface meshes -- the mesh's points are the matched MapPoints
(point vertices, FEA2::vVertices), the frame observes all of them, a few fixed keyframes observe subsets; observations carry
pixel noise and a few gross outliers; the frame's initial pose is off, and the surface is deformed between what the keyframes
saw and what the frame sees (the situation the FEM term exists for)."""
import numpy as np

CAM = (517.306408, 516.469215, 318.643040, 255.313989)      # TUM1-like pinhole


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _look_at(centre, eye):
    """World -> camera (R, t) of a camera at `eye` looking at `centre` (z forward)."""
    z = centre - eye; z /= np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.2]), z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    return R, -R @ eye


def _project(R, t, X, cam):
    c = X @ R.T + t
    return np.stack([c[:, 0] / c[:, 2] * cam[0] + cam[2], c[:, 1] / c[:, 2] * cam[1] + cam[3]], 1), c[:, 2]


def make_scene(top, seed=0, nkf=3, deform=0.03, pose_err=(0.02, 0.05), noise_px=0.7, outlier_frac=0.06, nlevels=8, scale=1.2):
    """top: [n, 3] float32 mesh points = initial MapPoint positions.  Returns a dict of flat arrays (the mini-g2o graph)."""
    rng = np.random.default_rng(seed)
    X0 = top.astype(np.float64)                                   # vPoint->setEstimate(toVector3d(pMP->GetWorldPos())), :586
    n = len(X0)
    centre = X0.mean(0); ext = np.linalg.norm(X0.max(0) - X0.min(0))
    # the TRUE surface the frame sees: the map's points moved by a smooth deformation field
    d = deform * ext * np.sin(2.0 * (X0 - centre) / ext + rng.uniform(0, 3, 3)) * rng.uniform(0.5, 1.0, 3)
    Xtrue = X0 + d
    eye = centre + np.array([0.1, -0.2, -2.2]) * ext
    Rf, tf = _look_at(centre, eye)
    kfR, kft = [], []
    for k in range(nkf):
        Rk, tk = _look_at(centre, eye + rng.normal(0, 0.35, 3) * ext)
        kfR.append(Rk); kft.append(tk)
    sigma2 = (scale ** np.arange(nlevels)) ** 2
    octave = rng.integers(0, 4, n)
    e_pt, e_cam, e_obs, e_info, e_K = [], [], [], [], []
    for i in range(n):                                           # Optimizer.cc:576-707: the frame edge, then the MP's keyframe edges
        uv, _ = _project(Rf, tf, Xtrue[i:i + 1], CAM)
        uv = uv[0] + rng.normal(0, noise_px * scale ** octave[i], 2)
        if rng.random() < outlier_frac:
            uv += rng.choice([-1, 1], 2) * rng.uniform(15, 60, 2)
        e_pt.append(i); e_cam.append(-1); e_obs.append(uv); e_info.append(1.0 / sigma2[octave[i]]); e_K.append(CAM)
        for k in range(nkf):
            if rng.random() < 0.6:
                uvk, _ = _project(kfR[k], kft[k], X0[i:i + 1], CAM)      # the keyframes saw the undeformed map
                uvk = uvk[0] + rng.normal(0, noise_px * scale ** octave[i], 2)
                if rng.random() < outlier_frac / 2:
                    uvk += rng.choice([-1, 1], 2) * rng.uniform(15, 60, 2)
                e_pt.append(i); e_cam.append(k); e_obs.append(uvk); e_info.append(1.0 / sigma2[octave[i]]); e_K.append(CAM)   # :672-674: the FRAME keypoint's octave
    R0 = _rodrigues(rng.normal(0, pose_err[0], 3)) @ Rf
    t0 = tf + rng.normal(0, pose_err[1], 3) * ext
    return {"R0": np.ascontiguousarray(R0), "t0": np.ascontiguousarray(t0), "kfR": np.ascontiguousarray(np.array(kfR).reshape(-1, 9)),
            "kft": np.ascontiguousarray(np.array(kft).reshape(-1, 3)), "X0": np.ascontiguousarray(X0),
            "e_pt": np.array(e_pt, np.int32), "e_cam": np.array(e_cam, np.int32), "e_obs": np.ascontiguousarray(np.array(e_obs, np.float64)),
            "e_info": np.array(e_info, np.float64), "e_K": np.ascontiguousarray(np.array(e_K, np.float64)), "Xtrue": Xtrue, "Rf": Rf, "tf": tf}


def write_scene(path, nElType, top, faces, derived, sc):
    with open(path, "wb") as f:
        f.write(np.array([nElType, len(top), len(faces), len(sc["X0"]), len(derived), len(sc["kfR"]), len(sc["e_pt"])], np.int32).tobytes())
        f.write(np.ascontiguousarray(top, np.float32).tobytes()); f.write(np.ascontiguousarray(faces, np.int32).tobytes())
        f.write(np.ascontiguousarray(derived, np.int32).tobytes())
        for k in ("R0", "t0", "kfR", "kft", "X0", "e_pt", "e_cam", "e_obs", "e_info", "e_K"):
            f.write(sc[k].tobytes())
