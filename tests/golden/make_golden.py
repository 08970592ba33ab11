#!/usr/bin/env python3
"""Generate the golden vectors G1..G8 by IMPORTING the reference in this container.

Run once, here (``python tests/golden/make_golden.py``); the produced ``*.npz`` / ``*.json`` files
are committed, the reference itself never is.  Nothing under tests/ reads /root/reference at test
time — only this generator does.

Third-party modules the reference imports that are absent offline:

* ``cv2`` (opencv-python~=4.5.2.54, /root/reference/requirements.txt:4).  bundleAdjuster.py touches
  it only through ``cv2.Rodrigues`` in the two result-formatting helpers
  (/root/reference/bundleAdjuster.py:153,201).  A placeholder module is seeded into ``sys.modules``
  whose ``Rodrigues`` is scipy's ``Rotation.from_rotvec(...).as_matrix()`` — an independent
  third-party implementation of the same closed-form map, NOT OpenCV.  Goldens that pass through it
  (the 4x4 / 3x4 extrinsics of G5/G6) therefore pin the *mathematical* Rodrigues map, and say so in
  their ``note`` field.  Everything else in G1-G6 never reaches cv2.
* ``pyntcloud`` (PLY writer; unused by the functions exercised) — empty placeholder.

OpenCV-bound functions (featureTracking, cv2.triangulatePoints, ORB) cannot run and have NO golden:
"parity unpinned" for those stages.
"""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np
import scipy
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

cv2_placeholder = types.ModuleType("cv2")
cv2_placeholder.Rodrigues = lambda rvec: (Rotation.from_rotvec(np.asarray(rvec, float).reshape(3)).as_matrix(), None)
sys.modules["cv2"] = cv2_placeholder
pc = types.ModuleType("pyntcloud")
pc.PyntCloud = object
sys.modules["pyntcloud"] = pc

import bundleAdjuster as ref_ba  # noqa: E402  (the reference)
import processor as ref_proc  # noqa: E402
from track import Track as RefTrack  # noqa: E402

from scipy.optimize import least_squares  # noqa: E402
from meatmodeler_amd import synth  # noqa: E402

VERS = dict(numpy=np.__version__, scipy=scipy.__version__)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def g1_rotate_project():
    rng = np.random.default_rng(101)
    n = 64
    pts = rng.normal(0, 3, (n, 3))
    rv = rng.normal(0, 1, (n, 3))
    rv[0] = 0.0                                   # theta = 0 row (nan_to_num branch)
    rv[1] = np.array([np.pi, 0, 0]) * (1 - 1e-9)  # |r| ~ pi
    rv[2] = np.array([0, 1e-9, 0])                # tiny angle
    rv[3] = np.array([2.0, -2.0, 1.0])            # theta = 3
    tv = rng.normal(0, 2, (n, 3)) + np.array([0, 0, 12.0])
    params = np.hstack([rv, tv])
    K = np.array([[1480.0, 1.5, 955.0], [0.0, 1510.0, 545.0], [0.0, 0.0, 1.0]])  # with skew
    rot = ref_ba.rotate(pts, rv)
    proj = ref_ba.project(pts, params, K)
    save("g1_rotate_project.npz", pts=pts, params=params, K=K, rotated=rot, projected=proj)


def g2_frame_parameters():
    rng = np.random.default_rng(202)
    F = 12
    ext34 = np.empty((F, 3, 4))
    for i in range(F):
        ext34[i, :, :3] = Rotation.from_rotvec(rng.normal(0, 0.8, 3)).as_matrix()
        ext34[i, :, 3] = rng.normal(0, 3, 3)
    ext34[0, :, :3] = np.eye(3)  # identity rotation -> theta = 0 -> nan_to_num branch
    ext44 = np.concatenate([ext34, np.tile(np.array([[[0, 0, 0, 1.0]]]), (F, 1, 1))], axis=1)
    with np.errstate(all="ignore"):
        p34 = ref_ba.frameParameters(ext34)
        p44 = ref_ba.frameParameters(ext44)
    save("g2_frame_parameters.npz", ext34=ext34, ext44=ext44, params34=p34, params44=p44)


def g3_point_pose_fun():
    out = {}
    for tag, (F, P, L, seed) in dict(small=(6, 40, 4, 3), mid=(40, 4000, 6, 1)).items():
        pr = synth.make_ba_problem(F, P, L, seed=seed)
        with np.errstate(all="ignore"):
            cams = ref_ba.frameParameters(pr["ext"])
        x = np.hstack([cams, pr["pts0"].reshape(-1)])
        res = ref_ba.pointFun(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
        sel = slice(None) if tag == "small" else slice(0, None, 37)
        out[f"{tag}_F"] = F
        out[f"{tag}_P"] = P
        out[f"{tag}_L"] = L
        out[f"{tag}_seed"] = seed
        out[f"{tag}_cams"] = cams
        out[f"{tag}_res"] = res.reshape(-1, 2)[sel]
        out[f"{tag}_sel"] = np.arange(res.size // 2)[sel]
        out[f"{tag}_cost"] = 0.5 * float(res @ res)
    # poseFun: 12 chessboard points per frame, 4 frames
    F = 4
    pr = synth.make_ba_problem(F, 12, 4, seed=9)
    pts3 = np.zeros((12, 3))
    grid = np.mgrid[0:4, 0:3].T.reshape(-1, 2) * 2
    pts3[:, 0] = grid[:, 0]
    pts3[:, 2] = grid[:, 1]
    fi = np.repeat(np.arange(F), 12)
    pi = np.tile(np.arange(12), F)
    with np.errstate(all="ignore"):
        cams = ref_ba.frameParameters(pr["ext"])
    obs = ref_ba.project(pts3[pi], cams.reshape(F, 6)[fi], pr["K"]) + 0.3
    out["pose_cams"] = cams
    out["pose_K"] = pr["K"]
    out["pose_pts3"] = pts3
    out["pose_obs"] = obs
    out["pose_res"] = ref_ba.poseFun(cams, pr["K"], F, fi, pi, pts3, obs)
    save("g3_point_pose_fun.npz", **out)


def g4_sparsity():
    pr = synth.make_ba_problem(5, 9, 3, seed=4)
    A = ref_ba.pointAdjustmentSparsity(5, 9, pr["fi"], pr["pi"]).tocsr()
    A.sort_indices()
    save("g4_sparsity.npz", fi=pr["fi"], pi=pr["pi"], indptr=A.indptr, indices=A.indices,
         shape=np.array(A.shape))


def _similarity_align(X, Y):
    """Least-squares similarity (the 7-DoF gauge of a free bundle adjustment) mapping X onto Y."""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, S, Vt = np.linalg.svd(Yc.T @ Xc)
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / (Xc ** 2).sum()
    return s * Xc @ R.T + my


def _run_adjust(pr):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        pts, exts = ref_ba.adjustPoints(pr["ext"], pr["K"], pr["pts0"][:, None, :], pr["obs"], pr["fi"], pr["pi"])
    return pts, np.array(exts), buf.getvalue()


def g5_adjust_points():
    meta = {}
    for tag, (F, P, L, seed) in dict(a=(6, 40, 4, 3), b=(12, 300, 5, 5), c=(40, 2000, 6, 1)).items():
        pr = synth.make_ba_problem(F, P, L, seed=seed)
        pts, exts, table = _run_adjust(pr)
        # the same call with the optimiser result kept, to record x / cost / nfev (reference settings)
        with np.errstate(all="ignore"):
            cams0 = ref_ba.frameParameters(pr["ext"])
        x0 = np.hstack([cams0, pr["pts0"].reshape(-1)])
        A = ref_ba.pointAdjustmentSparsity(F, P, pr["fi"], pr["pi"])
        args = (pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
        with contextlib.redirect_stdout(io.StringIO()):
            r1 = least_squares(ref_ba.pointFun, x0, jac_sparsity=A, verbose=2, x_scale="jac", ftol=1e-4,
                               method="trf", args=args)  # bundleAdjuster.py:180-192 verbatim settings
            # G5(ii): the SAME reference cost function and sparsity driven to the minimiser.  With the reference's
            # own settings (2-point differences, LSMR at its default 1e-6) a 400-evaluation run was still creeping
            # (round 1: status 0), so the inner solve is tightened (LSMR atol = btol = 1e-14) and the differences are
            # 3-point; the run then stops on ftol in 10-20 evaluations, and a second, differently configured run
            # (2-point differences) lands on the same minimiser modulo the 7-DoF gauge: recorded as
            # `tight_repro_aligned` (similarity-aligned max point difference / scene size, 4e-7..8e-7).
            tight = dict(jac_sparsity=A, verbose=0, x_scale="jac", ftol=1e-13, xtol=1e-13, gtol=1e-13, method="trf",
                         args=args, max_nfev=3000, tr_solver="lsmr", tr_options=dict(atol=1e-14, btol=1e-14))
            r2 = least_squares(ref_ba.pointFun, x0, jac="3-point", **tight)
            r3 = least_squares(ref_ba.pointFun, x0, jac="2-point", **tight)
        assert r2.status > 0 and r3.status > 0, (r2.status, r3.status)
        p2, p3 = r2.x[6 * F:].reshape(P, 3), r3.x[6 * F:].reshape(P, 3)
        repro = float(np.abs(_similarity_align(p3, p2) - p2).max() / np.abs(p2).max())
        assert repro < 1e-5, repro
        assert np.allclose(r1.x[6 * F:].reshape(P, 3), pts, rtol=0, atol=0)
        save(f"g5_adjust_points_{tag}.npz", F=F, P=P, L=L, seed=seed, x0=x0, points=pts, extrinsics=exts,
             x_ref=r1.x, cost_ref=r1.cost, nfev_ref=r1.nfev, njev_ref=r1.njev, optimality_ref=r1.optimality,
             status_ref=r1.status, x_tight=r2.x, cost_tight=r2.cost, nfev_tight=r2.nfev,
             status_tight=r2.status, optimality_tight=r2.optimality, tight_repro_aligned=repro,
             cost_tight_2pt=r3.cost, nfev_tight_2pt=r3.nfev, cost0=0.5 * float(r1.fun @ r1.fun) * 0 + 0.5 * float(
                 ref_ba.pointFun(x0, *args) @ ref_ba.pointFun(x0, *args)))
        meta[tag] = dict(table=table, nfev=int(r1.nfev), status=int(r1.status))
    meta["versions"] = VERS
    meta["note"] = ("extrinsics pass through a cv2.Rodrigues placeholder implemented with "
                    "scipy.spatial.transform.Rotation (OpenCV absent); x_ref/points do not.")
    with open(os.path.join(HERE, "g5_adjust_points_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def g5_adjust_points_large():
    """G5 case d: a problem wide enough for the two-ended banded factorisation (120 cameras -> 12 blocks of 64, band of
    one block), reference settings and converged; only every 10th point of the minimisers is stored (small fixture)."""
    F, P, L, seed = 120, 6000, 8, 21
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    with np.errstate(all="ignore"):
        cams0 = ref_ba.frameParameters(pr["ext"])
    x0 = np.hstack([cams0, pr["pts0"].reshape(-1)])
    A = ref_ba.pointAdjustmentSparsity(F, P, pr["fi"], pr["pi"])
    args = (pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
    with contextlib.redirect_stdout(io.StringIO()):
        r1 = least_squares(ref_ba.pointFun, x0, jac_sparsity=A, verbose=2, x_scale="jac", ftol=1e-4, method="trf", args=args)
        tight = dict(jac_sparsity=A, verbose=0, x_scale="jac", ftol=1e-13, xtol=1e-13, gtol=1e-13, method="trf",
                     args=args, max_nfev=3000, tr_solver="lsmr", tr_options=dict(atol=1e-14, btol=1e-14))
        r2 = least_squares(ref_ba.pointFun, x0, jac="3-point", **tight)
    assert r2.status > 0, r2.status
    sub = np.arange(0, P, 10)
    save("g5_adjust_points_d.npz", F=F, P=P, L=L, seed=seed, sub=sub, cost0=0.5 * float(np.sum(ref_ba.pointFun(x0, *args) ** 2)),
         cost_ref=r1.cost, nfev_ref=r1.nfev, status_ref=r1.status, points_ref_sub=r1.x[6 * F:].reshape(P, 3)[sub],
         cost_tight=r2.cost, nfev_tight=r2.nfev, status_tight=r2.status, optimality_tight=r2.optimality,
         points_tight_sub=r2.x[6 * F:].reshape(P, 3)[sub], cams_tight=r2.x[:6 * F])


def g6_adjust_pose():
    F = 5
    rng = np.random.default_rng(66)
    ext_gt = synth.orbit_cameras(F, arc_deg=40.0, radius=12.0, height=-3.0)
    K = synth.default_K(1920, 1080)
    pts3 = np.zeros((12, 3))
    grid = np.mgrid[0:4, 0:3].T.reshape(-1, 2) * 2
    pts3[:, 0] = grid[:, 0]
    pts3[:, 2] = grid[:, 1]
    obs = []
    for f in range(F):
        Xc = pts3 @ ext_gt[f, :, :3].T + ext_gt[f, :, 3]
        u = Xc @ K.T
        obs.append(u[:, :2] / u[:, 2:3])
    obs = np.concatenate(obs) + rng.normal(0, 0.2, (12 * F, 2))
    ext0 = ext_gt.copy()
    ext0[:, :, 3] += rng.normal(0, 0.05, (F, 3))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        out = ref_ba.adjustPose(ext0, K, obs)
    save("g6_adjust_pose.npz", ext0=ext0, K=K, obs=obs, result=np.array(out))
    with open(os.path.join(HERE, "g6_adjust_pose_table.txt"), "w") as fh:
        fh.write(buf.getvalue())


def _dump_tracks(tracks):
    return [dict(coords=[[int(k), [float(v[0]), float(v[1])]] for k, v in t.getCoordinates().items()],
                 updated=bool(t.wasUpdated())) for t in tracks]


def g7_point_tracking():
    """Scripted keyframe match sequences through the reference's pointTracking + managePoints."""
    rng = np.random.default_rng(77)
    scripts = []
    # script 0: hand-written edge cases (duplicate prev coordinates, many-to-one, death/rebirth)
    kp = {0: [(10.0, 10.0), (20.5, 11.25), (30.0, 12.0), (10.0, 10.0)],  # kp 3 duplicates kp 0's coordinates
          1: [(11.0, 10.0), (21.5, 11.25), (31.0, 12.0), (41.0, 13.0)],
          2: [(12.0, 10.0), (22.5, 11.25), (32.0, 12.0), (42.0, 13.0), (52.0, 14.0)],
          3: [(13.0, 10.0), (23.5, 11.25), (33.0, 12.0)],
          4: [(14.0, 10.0), (24.5, 11.25)]}
    matches = {0: [(0, 0), (1, 1), (2, 1), (3, 3)],   # many-to-one onto kp(1,1); duplicate-coordinate query 3
               1: [(0, 0), (1, 2), (3, 3), (2, 4)],   # out-of-order trains
               2: [(1, 1), (4, 2)],                   # tracks die; (2,4) continues
               3: [(0, 0), (1, 1), (2, 1)]}           # rebirth from frame 3
    scripts.append((kp, matches))
    # scripts 1,2: random sequences with collisions
    for s in range(2):
        nk, nf = 40 + 25 * s, 5
        kp = {f: [tuple(map(float, p)) for p in np.round(rng.uniform(0, 500, (nk, 2)) * 4) / 4] for f in range(nf)}
        for f in range(nf):      # force a few duplicate coordinates inside a frame
            kp[f][5] = kp[f][2]
            kp[f][17] = kp[f][9]
        matches = {}
        for f in range(nf - 1):
            q = np.sort(rng.choice(nk, size=int(nk * 0.7), replace=False))
            t = rng.integers(0, nk, size=q.size)  # with replacement -> many-to-one
            matches[f] = list(zip(q.tolist(), t.tolist()))
        scripts.append((kp, matches))
    out = []
    for kp, matches in scripts:
        tracks, popped_all, calls = [], [], []
        for f in sorted(matches):
            m = matches[f]
            prev = np.array([kp[f][q] for q, _ in m])
            cur = np.array([kp[f + 1][t] for _, t in m])
            popped, tracks = ref_proc.pointTracking(tracks, f, prev, f + 1, cur)
            popped_all += popped
            calls.append(dict(prev_ID=f, ID=f + 1, popped=_dump_tracks(popped), updated=_dump_tracks(tracks)))
        final = popped_all + tracks
        for i, t in enumerate(final):
            t.setPoint(np.array([[float(i), float(i) + 0.5, -float(i)]]))
        points, coords, fidx, pidx = ref_proc.managePoints(final)
        out.append(dict(kp={str(k): v for k, v in kp.items()}, matches={str(k): v for k, v in matches.items()},
                        calls=calls,
                        manage=dict(points_shape=list(np.array(points).shape),
                                    coordinates=[[float(c[0]), float(c[1])] for c in coords],
                                    frame_indices=[int(i) for i in fidx], point_indices=[int(i) for i in pidx])))
    with open(os.path.join(HERE, "g7_point_tracking.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote g7_point_tracking.json", len(out), "scripts")


def g8_track_api():
    t = RefTrack(3, (1.0, 2.0), 4, (1.5, 2.5))
    log = [dict(op="init", coords=_dump_tracks([t])[0], tri=list(t.getTriangulationData()), point=t.getPoint())]
    t.update(5, (2.0, 3.0))
    log.append(dict(op="update5", coords=_dump_tracks([t])[0], tri=list(t.getTriangulationData())))
    t.reset()
    log.append(dict(op="reset", updated=t.wasUpdated(), get4=list(t.getCoordinate(4)), get9=t.getCoordinate(9)))
    t.update(4, (9.0, 9.0))  # overwrite keeps dict position
    log.append(dict(op="update4", coords=_dump_tracks([t])[0], tri=list(t.getTriangulationData())))
    with open(os.path.join(HERE, "g8_track_api.json"), "w") as fh:
        json.dump(log, fh)
    print("wrote g8_track_api.json")


if __name__ == "__main__":
    todo = dict(g1=g1_rotate_project, g2=g2_frame_parameters, g3=g3_point_pose_fun, g4=g4_sparsity,
                g5=g5_adjust_points, g5d=g5_adjust_points_large, g6=g6_adjust_pose, g7=g7_point_tracking, g8=g8_track_api)
    for name in (sys.argv[1:] or list(todo)):          # `make_golden.py g5` regenerates one family
        todo[name]()
    with open(os.path.join(HERE, "VERSIONS.json"), "w") as fh:
        json.dump(VERS, fh)
