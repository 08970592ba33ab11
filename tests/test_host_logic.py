"""CPU: host-side logic of the product (track linking, flattening, index build, level geometry, trust-region
scalars) against the golden vectors captured from the reference and against the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from meatmodeler_amd import ops, processor, bundleAdjuster, synth, parallel
from meatmodeler_amd._lib import lib, c_i32p, c_i64p, c_f32p
from meatmodeler_amd.track import Track
from meatmodeler_amd.pipeline import ClipPipeline
from oracle import ba_oracle as bo
from oracle import orb_oracle as oo


def _dump(tracks):
    return [dict(coords=[[int(k), [float(v[0]), float(v[1])]] for k, v in t.getCoordinates().items()],
                 updated=bool(t.wasUpdated())) for t in tracks]


def _scripts(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g7_point_tracking.json")))


def test_point_tracking_and_manage_points_golden(golden_dir):
    """processor.pointTracking / managePoints reproduce the reference's outputs call by call (G7)."""
    for sc in _scripts(golden_dir):
        kp = {int(k): v for k, v in sc["kp"].items()}
        matches = {int(k): v for k, v in sc["matches"].items()}
        tracks, popped_all = [], []
        for call in sc["calls"]:
            f = call["prev_ID"]
            m = matches[f]
            prev = np.array([kp[f][q] for q, _ in m])
            cur = np.array([kp[f + 1][t] for _, t in m])
            popped, tracks = processor.pointTracking(tracks, f, prev, f + 1, cur)
            popped_all += popped
            assert _dump(popped) == call["popped"]
            assert _dump(tracks) == call["updated"]
        final = popped_all + tracks
        for i, t in enumerate(final):
            t.setPoint(np.array([[float(i), float(i) + 0.5, -float(i)]]))
        points, coords, fidx, pidx = processor.managePoints(final)
        mg = sc["manage"]
        assert list(np.array(points).shape) == mg["points_shape"]
        assert [[float(c[0]), float(c[1])] for c in coords] == mg["coordinates"]
        assert [int(i) for i in fidx] == mg["frame_indices"]
        assert [int(i) for i in pidx] == mg["point_indices"]


def test_link_tracks_clip_golden(golden_dir):
    """mm_link_tracks_clip (bulk C linker) == the reference's per-call pointTracking + managePoints (G7)."""
    for sc in _scripts(golden_dir):
        kp = {int(k): v for k, v in sc["kp"].items()}
        matches = {int(k): v for k, v in sc["matches"].items()}
        F = len(kp)
        cap = max(len(v) for v in kp.values())
        mcap = max(len(v) for v in matches.values())
        kp_xy = np.zeros((F, cap, 2), np.float32)
        kp_count = np.zeros(F, np.int32)
        for f, v in kp.items():
            kp_xy[f, :len(v)] = v
            kp_count[f] = len(v)
        mm = np.zeros((F - 1, mcap, 2), np.int32)
        mc = np.zeros(F - 1, np.int32)
        for f, v in matches.items():
            mm[f, :len(v)] = v
            mc[f] = len(v)
        tp, of, ok = ClipPipeline.link(None, kp_count, kp_xy, mc, mm)
        coords, fi, pi = ClipPipeline.flatten(tp, of, ok, kp_xy)
        mg = sc["manage"]
        assert coords.tolist() == mg["coordinates"]
        assert fi.tolist() == mg["frame_indices"]
        assert pi.tolist() == mg["point_indices"]
        assert len(tp) - 1 == mg["points_shape"][0]


def test_link_tracks_clip_random_vs_oracle():
    rng = np.random.default_rng(123)
    F, nk = 9, 120
    kp_xy = (np.round(rng.uniform(0, 300, (F, nk, 2)) * 2) / 2).astype(np.float32)
    kp_xy[:, 7] = kp_xy[:, 3]       # duplicate coordinates in every frame
    kp_count = np.full(F, nk, np.int32)
    kp_count[4] = 100
    mc = np.zeros(F - 1, np.int32)
    mm = np.zeros((F - 1, nk, 2), np.int32)
    for f in range(F - 1):
        q = np.sort(rng.choice(kp_count[f], size=int(kp_count[f] * 0.6), replace=False))
        t = rng.integers(0, kp_count[f + 1], size=q.size)
        mc[f] = q.size
        mm[f, :q.size, 0] = q
        mm[f, :q.size, 1] = t
    tp, of, ok = ClipPipeline.link(None, kp_count, kp_xy, mc, mm)
    tracks, popped = [], []
    for f in range(F - 1):
        m = mm[f, :mc[f]]
        p, tracks = bo.point_tracking(tracks, f, kp_xy[f][m[:, 0]].astype(np.float64), f + 1,
                                      kp_xy[f + 1][m[:, 1]].astype(np.float64))
        popped += p
    final = popped + tracks
    assert len(final) == len(tp) - 1
    _, coords, fidx, pidx = bo.manage_points(final)
    c2, f2, p2 = ClipPipeline.flatten(tp, of, ok, kp_xy)
    assert [[float(a), float(b)] for a, b in coords] == c2.tolist()
    assert list(fidx) == f2.tolist() and list(pidx) == p2.tolist()


def test_link_tracks_empty_and_single_frame():
    tp, of, ok = ClipPipeline.link(None, np.zeros(1, np.int32), np.zeros((1, 4, 2), np.float32), np.zeros(0, np.int32),
                                   np.zeros((0, 4, 2), np.int32))
    assert len(tp) == 1 and len(of) == 0
    tp, of, ok = ClipPipeline.link(None, np.array([3, 3], np.int32), np.zeros((2, 4, 2), np.float32),
                                   np.zeros(1, np.int32), np.zeros((1, 4, 2), np.int32))
    assert len(tp) == 1 and len(of) == 0


def test_track_api_golden(golden_dir):
    log = json.load(open(os.path.join(golden_dir, "g8_track_api.json")))
    t = Track(3, (1.0, 2.0), 4, (1.5, 2.5))
    tri = lambda: json.loads(json.dumps(list(t.getTriangulationData())))
    assert _dump([t])[0] == log[0]["coords"] and tri() == log[0]["tri"] and t.getPoint() is None
    t.update(5, (2.0, 3.0))
    assert t.wasUpdated() and _dump([t])[0] == log[1]["coords"] and tri() == log[1]["tri"]
    t.reset()
    assert t.wasUpdated() == log[2]["updated"] and list(t.getCoordinate(4)) == log[2]["get4"]
    assert t.getCoordinate(9) is None
    t.update(4, (9.0, 9.0))
    assert _dump([t])[0] == log[3]["coords"] and tri() == log[3]["tri"]


def test_frame_parameters_and_sparsity_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "g2_frame_parameters.npz"))
    np.testing.assert_allclose(bundleAdjuster.frameParameters(d["ext34"]), d["params34"], rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(bundleAdjuster.frameParameters(d["ext44"]), d["params44"], rtol=1e-13, atol=1e-14)
    s = np.load(os.path.join(golden_dir, "g4_sparsity.npz"))
    A = bundleAdjuster.pointAdjustmentSparsity(5, 9, s["fi"], s["pi"]).tocsr()
    A.sort_indices()
    np.testing.assert_array_equal(A.indptr, s["indptr"])
    np.testing.assert_array_equal(A.indices, s["indices"])


def test_reformat_results_match_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "g5_adjust_points_a.npz"))
    F, P = int(d["F"]), int(d["P"])

    class R:
        x = d["x_ref"]
    pts, ext = bundleAdjuster.reformatPointResult(R, F, P)
    np.testing.assert_array_equal(pts, d["points"])
    np.testing.assert_allclose(np.array(ext), d["extrinsics"], rtol=1e-12, atol=1e-13)
    g6 = np.load(os.path.join(golden_dir, "g6_adjust_pose.npz"))
    cams = np.array([np.concatenate([bo.frame_parameters(e[None])[:3], e[:, 3]]) for e in g6["result"]])

    class R2:
        x = cams.ravel()
    out = bundleAdjuster.reformatPoseResult(R2, len(cams))
    np.testing.assert_allclose(np.array(out), g6["result"], rtol=1e-9, atol=1e-10)


def test_ba_build_index_stable_csr():
    rng = np.random.default_rng(1)
    F, P, O = 7, 40, 500
    fi = rng.integers(0, F, O).astype(np.int32)
    pi = rng.integers(0, P, O).astype(np.int32)
    pt_ptr, pt_obs, cam_ptr, cam_obs = ops.ba_build_index(F, P, fi, pi)
    np.testing.assert_array_equal(pt_obs, np.argsort(pi, kind="stable"))
    np.testing.assert_array_equal(cam_obs, np.argsort(fi, kind="stable"))
    np.testing.assert_array_equal(np.diff(pt_ptr), np.bincount(pi, minlength=P))
    np.testing.assert_array_equal(np.diff(cam_ptr), np.bincount(fi, minlength=F))
    with pytest.raises(ValueError):
        ops.ba_build_index(F, P, np.array([F], np.int32), np.array([0], np.int32))


def test_ba_build_pairs_matches_brute_force():
    import ctypes as C
    rng = np.random.default_rng(3)
    F, P, O = 9, 30, 160
    fi = rng.integers(0, F, O).astype(np.int32)          # arbitrary order, repeated (camera, point) pairs included
    pi = rng.integers(0, P, O).astype(np.int32)
    pt_ptr, pt_obs, cam_ptr, cam_obs = ops.ba_build_index(F, P, fi, pi)
    span = F - 1
    seg_ptr = np.zeros(F * (span + 1) + 1, np.int64)
    i32p, i64p = c_i32p, c_i64p
    args = (F, P, O, fi.ctypes.data_as(i32p), pi.ctypes.data_as(i32p), pt_ptr.ctypes.data_as(i32p),
            np.ascontiguousarray(pt_obs).ctypes.data_as(i32p), cam_ptr.ctypes.data_as(i32p),
            np.ascontiguousarray(cam_obs).ctypes.data_as(i32p), span, seg_ptr.ctypes.data_as(i64p))
    n = lib.mm_ba_build_pairs(*args, None, None, 0)
    want = [(int(fi[o]) * (span + 1) + int(fi[o] - fi[o2]), o, o2) for o in range(O) for o2 in range(O)
            if pi[o] == pi[o2] and fi[o2] <= fi[o]]
    assert n == len(want)
    po, po2 = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert lib.mm_ba_build_pairs(*args, po.ctypes.data_as(i32p), po2.ctypes.data_as(i32p), n) == n
    got = []
    for s_ in range(F * (span + 1)):
        for e in range(seg_ptr[s_], seg_ptr[s_ + 1]):
            got.append((s_, int(po[e]), int(po2[e])))
    assert sorted(got) == sorted(want)
    # canonical order inside a segment: camera-CSR order of o, then point-CSR order of o2
    rank_cam = np.empty(O, np.int64)
    rank_cam[cam_obs] = np.arange(O)
    rank_pt = np.empty(O, np.int64)
    rank_pt[pt_obs] = np.arange(O)
    for s_ in range(F * (span + 1)):
        seg = [(rank_cam[po[e]], rank_pt[po2[e]]) for e in range(seg_ptr[s_], seg_ptr[s_ + 1])]
        assert seg == sorted(seg)
    assert lib.mm_ba_build_pairs(*args[:9], 2, seg_ptr.ctypes.data_as(i64p), None, None, 0) == -1   # span too small


@pytest.mark.parametrize("hw,nf", [((1080, 1920), 4000), ((480, 640), 1000), ((2160, 3840), 8000), ((364, 652), 300),
                                   ((1080, 1920), 20000)])
def test_orb_level_geometry_matches_oracle(hw, nf):
    prm = ops.orb_params(nf)
    w, h, n, s = ops.orb_level_sizes(hw[0], hw[1], prm)
    wo, ho, no, so = oo.level_sizes(hw[0], hw[1], nf)
    np.testing.assert_array_equal(w, wo)
    np.testing.assert_array_equal(h, ho)
    np.testing.assert_array_equal(n, no)
    np.testing.assert_array_equal(s, so)
    assert n.sum() == nf


def test_trust_region_scalar_helpers_match_scipy():
    from scipy.optimize._lsq import common
    rng = np.random.default_rng(0)
    for _ in range(50):
        M = rng.normal(size=(2, 2))
        B = M @ M.T + 1e-3 * np.eye(2)
        g = rng.normal(size=2) * 10
        for Delta in (1e-3, 0.5, 10.0):
            p, nw = bundleAdjuster._solve_trust_region_2d(B, g, Delta)
            ps, nws = common.solve_trust_region_2d(B, g, Delta)
            np.testing.assert_allclose(p, ps, rtol=1e-12, atol=1e-14)
            assert nw == nws
    assert bundleAdjuster._update_tr_radius(1.0, 0.1, 1.0, 0.5, True) == common.update_tr_radius(1.0, 0.1, 1.0, 0.5, True)
    assert bundleAdjuster._update_tr_radius(1.0, 0.9, 1.0, 0.99, True) == common.update_tr_radius(1.0, 0.9, 1.0, 0.99, True)
    for args in [(1e-6, 1.0, 1e-3, 1.0, 0.5, 1e-4, 1e-8), (1e-2, 1.0, 1e-12, 1.0, 0.5, 1e-4, 1e-8),
                 (1e-6, 1.0, 1e-12, 1.0, 0.5, 1e-4, 1e-8), (1e-6, 1.0, 1e-3, 1.0, 0.1, 1e-4, 1e-8)]:
        assert bundleAdjuster._check_termination(*args) == common.check_termination(*args)


def test_partitions_cover_everything_once():
    for n, world in [(499, 8), (3, 8), (0, 2), (17, 4)]:
        seen = []
        for r in range(world):
            lo, hi = parallel.block_range(n, r, world)
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    pr = synth.make_ba_problem(20, 333, 5, seed=2)
    total = 0
    last = 0
    for r in range(4):
        lo, hi, mask = parallel.partition_points(pr["fi"], pr["pi"], 333, r, 4)
        assert lo == last
        last = hi
        total += mask.sum()
        assert abs(mask.sum() - len(pr["fi"]) / 4) <= 10
    assert last == 333 and total == len(pr["fi"])
    (p_lo, p_hi), (f_lo, f_hi) = parallel.pair_block(500, 7, 8)
    assert p_hi == 499 and f_hi == 500 and f_lo == p_lo


def test_brief_pattern_is_fixed_and_inside_the_disc():
    from meatmodeler_amd.orb_pattern import brief_pattern
    p = brief_pattern().astype(int)
    assert p.shape == (256, 4)
    assert ((p[:, 0] ** 2 + p[:, 1] ** 2) <= 225).all() and ((p[:, 2] ** 2 + p[:, 3] ** 2) <= 225).all()
    assert not ((p[:, 0] == p[:, 2]) & (p[:, 1] == p[:, 3])).any()
    assert int(p.sum()) == int(brief_pattern().astype(int).sum())
    import hashlib
    assert hashlib.sha1(brief_pattern().tobytes()).hexdigest() == PATTERN_SHA1


PATTERN_SHA1 = "12fa8b026f52705ed79d4c58f8d8ff7fe21aba5a"


def test_exact_trust_region_on_block_diagonal_matches_scipy():
    """adjustPose's trust-region solve on per-camera 6x6 eigen-decompositions == SciPy's SVD-based
    solve_lsq_trust_region on the dense block-diagonal Jacobian."""
    from scipy.optimize._lsq.common import solve_lsq_trust_region
    from scipy.linalg import svd
    rng = np.random.default_rng(4)
    F, rows = 5, 24
    blocks = rng.normal(size=(F, rows, 6)) * rng.uniform(0.1, 30, size=(F, 1, 6))
    J = np.zeros((F * rows, F * 6))
    for f in range(F):
        J[f * rows:(f + 1) * rows, f * 6:(f + 1) * 6] = blocks[f]
    r = rng.normal(size=F * rows) * 5
    U, s, Vt = svd(J, full_matrices=False)
    uf = U.T @ r
    B = np.einsum("fri,frj->fij", blocks, blocks)
    g = np.einsum("fri,fr->fi", blocks, r.reshape(F, rows))
    lam, V = np.linalg.eigh(B)
    vg = np.einsum("fji,fj->fi", V, g)
    for Delta in (1e-3, 0.05, 0.5, 1e3):
        for a0 in (0.0, 0.7):
            p_ref, alpha_ref, it_ref = solve_lsq_trust_region(F * 6, F * rows, uf, s, Vt.T, Delta, initial_alpha=a0)
            p, alpha, it = bundleAdjuster._solve_lsq_trust_region_eig(lam, vg, V, Delta, F * rows, a0)
            np.testing.assert_allclose(p.ravel(), p_ref, rtol=1e-9, atol=1e-12)
            assert it == it_ref and abs(alpha - alpha_ref) <= 1e-9 * max(1.0, abs(alpha_ref))


def test_save_point_cloud_writes_binary_ply(tmp_path):
    """processor.py:480-485: PyntCloud(DataFrame(points, x / y / z)).to_file(path + "Cloud.ply") -> binary PLY, doubles."""
    from meatmodeler_amd import processor
    pts = np.random.default_rng(0).normal(size=(37, 3))
    name = processor.savePointCloud(pts, str(tmp_path) + "/scan")
    assert name.endswith("scanCloud.ply")
    raw = open(name, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    assert lines[3:6] == ["property double x", "property double y", "property double z"]
    np.testing.assert_array_equal(np.frombuffer(body, "<f8").reshape(37, 3), pts)
    assert processor.savePointCloud(np.zeros((0, 3)), str(tmp_path) + "/empty").endswith("emptyCloud.ply")


def test_frame_oracle_definitions():
    """The CPU definitions of the per-frame front end (oracle/frame_oracle.c): LK recovers a known translation, the
    fixed-point LAB conversion is as accurate as 8-bit LAB allows, CLAHE leaves a constant image constant, corners
    respect the minimum distance."""
    from oracle import frame_oracle as fo
    from meatmodeler_amd import frame_tables, synth
    frames, _, _ = synth.render_orbit_frames(1, 320, 240, arc_deg=1.0, seed=2)
    a = frames[0]
    c = fo.good_features(a, 60, 0.05, 12, 5)
    d = np.linalg.norm(c[:, None] - c[None], axis=2) + 1e9 * np.eye(len(c))
    assert len(c) > 10 and d.min() >= 12
    c = c[(c[:, 0] > 30) & (c[:, 0] < 290) & (c[:, 1] > 30) & (c[:, 1] < 210)]
    nx, st, er = fo.lk_track(a, np.roll(np.roll(a, 1, 0), -2, 1), c, (21, 21), 2, 30, 0.01)
    assert st.mean() > 0.9 and np.abs(np.median((nx - c)[st == 1], 0) - [-2, 1]).max() < 0.05
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (4000, 1, 3), dtype=np.uint8)
    bgr[:2, 0] = [[0, 0, 0], [255, 255, 255]]
    lab, back = fo.lab_roundtrip(bgr, frame_tables.lab_tables())
    assert list(lab[0, 0]) == [0, 128, 128] and list(lab[1, 0]) == [255, 128, 128]
    err = np.abs(back.astype(int) - bgr.astype(int))
    assert (back[:2] == bgr[:2]).all() and err.mean() < 1.0
    flat = np.full((64, 96), 100, np.uint8)
    out = fo.clahe(flat)
    assert (out == out[0, 0]).all()


def test_bench_launcher_refuses_without_enough_gpus():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks -- and says so loudly when the box has
    fewer than N GPUs instead of reporting a one-GPU run as N.  On this CPU-only container: no GPU at
    all, so both the nccl and the gloo-rehearsal form refuse before any rank is started."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "MM_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--frames", "8", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, cwd=root)
    assert p.returncode != 0
    assert b"GPU(s) visible" in p.stderr or b"no GPU visible" in p.stderr
    assert not any(ln.startswith(b"{") for ln in p.stdout.splitlines())
