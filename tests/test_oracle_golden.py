"""Pin the CPU oracle (oracle/ba_oracle.py) against golden vectors captured from the reference's own
NumPy/SciPy code (tests/golden/make_golden.py).  CPU only."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from oracle import ba_oracle as bo
from meatmodeler_amd import synth


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_rotate_project(golden_dir):
    d = g(golden_dir, "g1_rotate_project.npz")
    np.testing.assert_allclose(bo.rotate(d["pts"], d["params"][:, :3]), d["rotated"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(bo.project(d["pts"], d["params"], d["K"]), d["projected"], rtol=1e-12, atol=1e-10)


def test_g2_frame_parameters(golden_dir):
    d = g(golden_dir, "g2_frame_parameters.npz")
    np.testing.assert_allclose(bo.frame_parameters(d["ext34"]), d["params34"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(bo.frame_parameters(d["ext44"]), d["params44"], rtol=1e-13, atol=1e-14)
    assert np.all(d["params34"][:3] == 0.0)  # identity rotation row -> zeros (nan_to_num)


def test_g3_point_and_pose_fun(golden_dir):
    d = g(golden_dir, "g3_point_pose_fun.npz")
    for tag in ("small", "mid"):
        F, P, L, seed = (int(d[f"{tag}_{k}"]) for k in ("F", "P", "L", "seed"))
        pr = synth.make_ba_problem(F, P, L, seed=seed)
        cams = bo.frame_parameters(pr["ext"])
        np.testing.assert_allclose(cams, d[f"{tag}_cams"], rtol=1e-13, atol=1e-14)
        x = np.hstack([cams, pr["pts0"].ravel()])
        res = bo.point_fun(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
        np.testing.assert_allclose(res.reshape(-1, 2)[d[f"{tag}_sel"]], d[f"{tag}_res"], rtol=1e-10, atol=1e-9)
        assert abs(0.5 * res @ res - float(d[f"{tag}_cost"])) <= 1e-12 * float(d[f"{tag}_cost"])
    F = 4
    fi = np.repeat(np.arange(F), 12)
    pi = np.tile(np.arange(12), F)
    np.testing.assert_allclose(d["pose_pts3"], bo.chessboard_points(12))
    res = bo.pose_fun(d["pose_cams"], d["pose_K"], F, fi, pi, d["pose_pts3"], d["pose_obs"])
    np.testing.assert_allclose(res, d["pose_res"], rtol=1e-10, atol=1e-9)


def test_g4_sparsity(golden_dir):
    d = g(golden_dir, "g4_sparsity.npz")
    A = bo.sparsity_pattern(5, 9, d["fi"], d["pi"]).tocsr()
    A.sort_indices()
    assert tuple(d["shape"]) == A.shape
    np.testing.assert_array_equal(A.indptr, d["indptr"])
    np.testing.assert_array_equal(A.indices, d["indices"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g5_adjust_points(golden_dir, tag):
    d = g(golden_dir, f"g5_adjust_points_{tag}.npz")
    meta = json.load(open(os.path.join(golden_dir, "g5_adjust_points_meta.json")))
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pts, ext, res = bo.adjust_points(pr["ext"], pr["K"], pr["pts0"][:, None, :], pr["obs"], pr["fi"], pr["pi"],
                                         verbose=2, return_result=True)
    # Same SciPy, same algorithm, same inputs — yet x only agrees to ~1e-3: the problem has a 7-DoF gauge
    # freedom (no camera is fixed, bundleAdjuster.py:160-194), the 2-point finite-difference Jacobian
    # amplifies last-bit differences of the residual function by 1/h ~ 7e7, and LSMR stops at 1e-6, so the
    # reference's own iterates are reproducible only to this level (measured: |dx| up to 1.9e-3 on case a,
    # 6e-5 on b/c; cost agrees to 4e-8).  The well-posed parity quantity is the cost / reprojection error.
    assert res.nfev == int(d["nfev_ref"]) and res.status == int(d["status_ref"])
    np.testing.assert_allclose(res.x, d["x_ref"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(pts, d["points"], rtol=0, atol=5e-3)
    np.testing.assert_allclose(np.array(ext), d["extrinsics"], rtol=0, atol=5e-3)
    assert abs(res.cost - float(d["cost_ref"])) <= 1e-6 * float(d["cost_ref"])
    assert buf.getvalue().splitlines()[0] == meta[tag]["table"].splitlines()[0]


def _similarity_align(X, Y):
    """Least-squares similarity (the 7-DoF gauge of a free bundle adjustment) mapping X onto Y."""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, S, Vt = np.linalg.svd(Yc.T @ Xc)
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / (Xc ** 2).sum()
    return s * Xc @ R.T + my


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g5_tight_minimiser_is_pinned(golden_dir, tag):
    """G5(ii): the reference's pointFun + sparsity driven to the minimiser (make_golden.py: LSMR at 1e-14, 3-point
    differences, ftol = xtol = gtol = 1e-13).  The run converged (status > 0), a second differently configured run of the
    same reference function reached the same minimiser modulo gauge to < 1e-6 of the scene size, and the oracle's cost
    function reproduces the recorded cost at x_tight; the minimum is below the reference-settings cost."""
    d = g(golden_dir, f"g5_adjust_points_{tag}.npz")
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    assert int(d["status_tight"]) > 0 and float(d["tight_repro_aligned"]) < 1e-6
    r = bo.point_fun(d["x_tight"], pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
    ct = float(d["cost_tight"])
    assert abs(0.5 * r @ r - ct) <= 1e-12 * ct
    assert ct <= float(d["cost_ref"]) and abs(float(d["cost_tight_2pt"]) - ct) <= 1e-10 * ct


def test_g5_case_d_fixture_is_converged(golden_dir):
    d = g(golden_dir, "g5_adjust_points_d.npz")
    assert int(d["status_tight"]) > 0 and int(d["status_ref"]) == 2
    assert float(d["cost_tight"]) <= float(d["cost_ref"]) < float(d["cost0"])
    assert d["points_tight_sub"].shape == (len(d["sub"]), 3)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g5_tight_driver_reaches_reference_minimiser(golden_dir, tag):
    """The product's trust-region driver (bundleAdjuster.SchurTRF) run to ftol = xtol = gtol = 1e-12 on a NumPy
    stand-in for the HIP sweeps (oracle cost function, exact Schur solve): 3-D points within 1e-4 relative of the
    reference minimiser after removing the gauge (north star), cost within 1e-8.  The same assertion runs against the
    HIP kernels in tests/test_gpu_parity.py::test_adjust_points_tight_vs_reference_minimiser."""
    import torch
    from test_distributed_cpu import NumpyBA
    from meatmodeler_amd.bundleAdjuster import SchurTRF
    d = g(golden_dir, f"g5_adjust_points_{tag}.npz")
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    pb = NumpyBA(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P)
    res = SchurTRF(pb).solve(torch.from_numpy(bo.frame_parameters(pr["ext"]).reshape(F, 6)),
                             torch.from_numpy(pr["pts0"].copy()), ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=100)
    ref = d["x_tight"][6 * F:].reshape(P, 3)
    scale = np.abs(ref).max()
    ct = float(d["cost_tight"])
    assert abs(res.cost - ct) <= 1e-8 * ct, (res.cost, ct)
    err = np.abs(_similarity_align(res.pts.numpy(), ref) - ref).max() / scale
    assert err <= 1e-4, err


def test_g6_adjust_pose(golden_dir):
    d = g(golden_dir, "g6_adjust_pose.npz")
    out = bo.adjust_pose(d["ext0"], d["K"], d["obs"])
    np.testing.assert_allclose(np.array(out), d["result"], rtol=1e-8, atol=1e-9)


def _dump(tracks):
    return [dict(coords=[[int(k), [float(v[0]), float(v[1])]] for k, v in t.getCoordinates().items()],
                 updated=bool(t.wasUpdated())) for t in tracks]


def test_g7_point_tracking_and_manage_points(golden_dir):
    scripts = json.load(open(os.path.join(golden_dir, "g7_point_tracking.json")))
    for sc in scripts:
        kp = {int(k): v for k, v in sc["kp"].items()}
        matches = {int(k): v for k, v in sc["matches"].items()}
        tracks, popped_all = [], []
        for call in sc["calls"]:
            f = call["prev_ID"]
            m = matches[f]
            prev = np.array([kp[f][q] for q, _ in m])
            cur = np.array([kp[f + 1][t] for _, t in m])
            popped, tracks = bo.point_tracking(tracks, f, prev, f + 1, cur)
            popped_all += popped
            assert _dump(popped) == call["popped"]
            assert _dump(tracks) == call["updated"]
        final = popped_all + tracks
        for i, t in enumerate(final):
            t.setPoint(np.array([[float(i), float(i) + 0.5, -float(i)]]))
        points, coords, fidx, pidx = bo.manage_points(final)
        mg = sc["manage"]
        assert list(np.array(points).shape) == mg["points_shape"]
        assert [[float(c[0]), float(c[1])] for c in coords] == mg["coordinates"]
        assert [int(i) for i in fidx] == mg["frame_indices"]
        assert [int(i) for i in pidx] == mg["point_indices"]


def test_g8_track_api(golden_dir):
    log = json.load(open(os.path.join(golden_dir, "g8_track_api.json")))
    t = bo.Track(3, (1.0, 2.0), 4, (1.5, 2.5))
    tri = lambda: json.loads(json.dumps(list(t.getTriangulationData())))
    assert _dump([t])[0] == log[0]["coords"] and tri() == log[0]["tri"] and t.getPoint() is None
    t.update(5, (2.0, 3.0))
    assert _dump([t])[0] == log[1]["coords"] and tri() == log[1]["tri"]
    t.reset()
    assert t.wasUpdated() == log[2]["updated"] and list(t.getCoordinate(4)) == log[2]["get4"]
    assert t.getCoordinate(9) is None
    t.update(4, (9.0, 9.0))
    assert _dump([t])[0] == log[3]["coords"] and tri() == log[3]["tri"]


def test_analytic_vs_fd_helper_consistency():
    """jacobian_fd (the checker for the HIP analytic Jacobian) agrees with point_fun differences."""
    pr = synth.make_ba_problem(5, 20, 3, seed=2)
    x = np.hstack([bo.frame_parameters(pr["ext"]), pr["pts0"].ravel()])
    Jc, Jp = bo.jacobian_fd(x, pr["K"], 5, 20, pr["fi"], pr["pi"], pr["obs"])
    d = np.zeros_like(x)
    d[6 * 5 + 3 * 7 + 1] = 1e-6
    num = (bo.point_fun(x + d, pr["K"], 5, 20, pr["fi"], pr["pi"], pr["obs"])
           - bo.point_fun(x - d, pr["K"], 5, 20, pr["fi"], pr["pi"], pr["obs"])) / 2e-6
    rows = np.where(pr["pi"] == 7)[0]
    np.testing.assert_allclose(num.reshape(-1, 2)[rows], Jp[rows, :, 1], rtol=1e-6, atol=1e-6)
