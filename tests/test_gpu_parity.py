"""GPU parity tests: every HIP path, called through the C ABI (ctypes), against the CPU oracle on identical seeded
inputs.  Bit-exact for integer / byte / index work; floating-point tolerances are written next to each check.

Run on the MI355X box:  python -m pytest tests -m gpu -q
"""
import contextlib
import io
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from meatmodeler_amd import ops, synth, processor, bundleAdjuster  # noqa: E402
from meatmodeler_amd._lib import default_context  # noqa: E402
from meatmodeler_amd.orb_pattern import brief_pattern  # noqa: E402
from meatmodeler_amd.pipeline import ClipPipeline  # noqa: E402
from oracle import ba_oracle as bo  # noqa: E402
from oracle import orb_oracle as oo  # noqa: E402

DEV = torch.device("cuda", 0)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


# ============================================================================================== matching

@pytest.mark.parametrize("n,seed", [(2000, 0), (4000, 1), (777, 2), (8000, 3)])     # C2, C3 and C5 key point counts
def test_bf_knn2_single_pair_bit_exact(n, seed):
    q, t, _ = synth.random_descriptors(n, seed=seed)
    t[5] = t[4]            # exact duplicate train rows -> distance ties
    q[9] = q[8]
    idx_o, dist_o = oo.bf_knn2(q, t)
    idx, dist = ops.bf_knn2(dev(q), dev(t))
    np.testing.assert_array_equal(dist.cpu().numpy(), dist_o)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_o)
    pairs, m = ops.ratio_filter_batched(idx.unsqueeze(0), dist.unsqueeze(0), 0.75)
    po = oo.ratio_filter(idx_o, dist_o, 0.75)
    assert int(m[0]) == len(po)
    np.testing.assert_array_equal(pairs[0, :len(po)].cpu().numpy(), po)


@pytest.mark.parametrize("variant", ["114", "200", "300", "310", "314"])
def test_bf_knn2_formulations_agree_with_oracle(variant, monkeypatch):
    """The three formulations of the Hamming search in csrc/bf_match.hip -- xor / popcount on the vector unit (114), the
    +1 / -1 GEMM on the int8 matrix instruction (200) and on the FP4 one (300, the default) -- against the C oracle on a
    ragged batch: per-pair counts that end inside a 32-train tile, an empty train set, duplicates (ties -> lowest train
    index), and a pair with more than 128 tiles (the 7-bit tile number of the packed keys wraps: segment folding)."""
    monkeypatch.setenv("MM_BF_VARIANT", variant)
    rng = np.random.default_rng(11)
    n_pairs, cap = 4, 4300
    q = rng.integers(0, 256, (n_pairs, cap, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (n_pairs, cap, 32), dtype=np.uint8)
    t[0, 70:80] = t[0, 3]                  # ten copies of one train row
    q[0, 5] = t[0, 3]
    t[3, 4200] = q[3, 17]                  # an exact match beyond tile 128
    nq = np.array([300, 1, 257, 4300], np.int32)
    nt = np.array([1000, 77, 0, 4241], np.int32)      # 1000 = 31 tiles + 8, 77 = 2 tiles + 13, empty, 132 tiles + 17
    idx, dist = ops.bf_knn2_batched(dev(q), dev(t), dev(nq), dev(nt))
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    for p in range(n_pairs):
        io, do = oo.bf_knn2(q[p, :nq[p]], t[p, :nt[p]])
        np.testing.assert_array_equal(dist[p, :nq[p]], do)
        np.testing.assert_array_equal(idx[p, :nq[p]], io)
    assert list(idx[0, 5]) == [3, 70] and list(dist[0, 5]) == [0, 0]
    assert idx[3, 17, 0] == 4200 and dist[3, 17, 0] == 0
    assert (idx[2, :257] == -1).all() and (dist[2, :257] == -1).all()


def test_bf_knn2_ties_resolve_to_lowest_train_index():
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    t[100:110] = t[7]                      # ten copies of row 7
    q = t[[7, 100, 250]].copy()
    idx, dist = ops.bf_knn2(dev(q), dev(t))
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert list(idx[0]) == [7, 100] and list(dist[0]) == [0, 0]
    assert list(idx[1]) == [7, 100]
    io_, do_ = oo.bf_knn2(q, t)
    np.testing.assert_array_equal(idx, io_)
    np.testing.assert_array_equal(dist, do_)


@pytest.mark.parametrize("nq,nt", [(0, 10), (5, 0), (5, 1), (5, 2), (1, 3), (513, 129)])
def test_bf_knn2_edge_sizes(nq, nt):
    rng = np.random.default_rng(nq * 31 + nt)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    idx, dist = ops.bf_knn2(dev(q), dev(t) if nt else torch.zeros((0, 32), dtype=torch.uint8, device=DEV))
    io_, do_ = oo.bf_knn2(q, t)
    np.testing.assert_array_equal(idx.cpu().numpy().reshape(-1, 2), io_)
    np.testing.assert_array_equal(dist.cpu().numpy().reshape(-1, 2), do_)
    if nq:
        pairs, m = ops.ratio_filter_batched(idx.unsqueeze(0), dist.unsqueeze(0), 0.75)
        po = oo.ratio_filter(io_, do_, 0.75)
        assert int(m[0]) == len(po)          # fewer than two neighbours -> never a match (processor.py:137)


def test_bf_knn2_batched_ragged_counts():
    rng = np.random.default_rng(11)
    F, cap = 7, 600
    table = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    counts = np.array([600, 512, 0, 1, 333, 600, 2], np.int32)
    for f in range(1, F):                      # plant matches between consecutive frames
        k = min(counts[f - 1], counts[f]) // 2
        table[f, :k] = table[f - 1, :k]
        flips = rng.random((k, 256)) < 0.03
        table[f, :k] ^= np.packbits(flips, axis=1, bitorder="little")
    d = dev(table)
    n = dev(counts)
    idx, dist = ops.bf_knn2_batched(d[:-1], d[1:], n[:-1].contiguous(), n[1:].contiguous())
    pairs, m = ops.ratio_filter_batched(idx, dist, 0.75, n[:-1].contiguous())
    idx, dist, pairs, m = idx.cpu().numpy(), dist.cpu().numpy(), pairs.cpu().numpy(), m.cpu().numpy()
    for p in range(F - 1):
        nq, nt = counts[p], counts[p + 1]
        io_, do_ = oo.bf_knn2(table[p, :nq], table[p + 1, :nt])
        np.testing.assert_array_equal(idx[p, :nq], io_, err_msg=f"pair {p}")
        np.testing.assert_array_equal(dist[p, :nq], do_, err_msg=f"pair {p}")
        po = oo.ratio_filter(io_, do_, 0.75)
        assert m[p] == len(po)
        np.testing.assert_array_equal(pairs[p, :len(po)], po)


def test_bf_knn2_full_size_properties():
    """C3 size (4000 x 4000) x 16 pairs: self-match property instead of the oracle (size independent)."""
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (17, 4000, 32), dtype=np.uint8)
    d = dev(base)
    idx, dist = ops.bf_knn2_batched(d[:-1], d[:-1])          # every set against itself
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    assert (dist[:, :, 0] == 0).all() and (idx[:, :, 0] == np.arange(4000)[None]).all()
    assert (dist[:, :, 1] > 0).all()
    sub = rng.choice(4000, 50, replace=False)
    io_, do_ = oo.bf_knn2(base[3, sub], base[3])
    np.testing.assert_array_equal(idx[3, sub], io_)
    np.testing.assert_array_equal(dist[3, sub], do_)


# ============================================================================================== ORB

def _orb_gpu(frames, nfeatures):
    B, H, W = frames.shape
    prm = ops.orb_params(nfeatures)
    wsp = ops.OrbWorkspace(B, H, W, prm, DEV, brief_pattern())
    xy, meta, resp, mom, desc, n = ops.orb_detect_compute(dev(frames), wsp)
    torch.cuda.synchronize()
    return [a.cpu().numpy() for a in (xy, meta, resp, mom, desc, n)]


def _cmp_orb(frames, nfeatures):
    xy, meta, resp, mom, desc, n = _orb_gpu(frames, nfeatures)
    for b in range(frames.shape[0]):
        r = oo.detect_compute(frames[b], nfeatures, brief_pattern())
        assert n[b] == r["n"], f"frame {b}: keypoint count {n[b]} vs oracle {r['n']}"
        k = r["n"]
        assert k > nfeatures // 4, "synthetic frame should be corner-rich"
        np.testing.assert_array_equal(meta[b, :k, :3], r["meta"][:, :3], err_msg=f"frame {b} (level,x,y)")
        np.testing.assert_array_equal(meta[b, :k, 3], r["meta"][:, 3], err_msg=f"frame {b} harris low bits")
        np.testing.assert_array_equal(xy[b, :k], r["xy"], err_msg=f"frame {b} pt")
        np.testing.assert_array_equal(resp[b, :k], r["resp"], err_msg=f"frame {b} response")
        np.testing.assert_array_equal(mom[b, :k], r["mom"], err_msg=f"frame {b} moments")
        bad = np.nonzero((desc[b, :k] != r["desc"]).any(axis=1))[0]
        assert bad.size == 0, f"frame {b}: {bad.size} descriptors differ, first {bad[:5]}"


def test_orb_small_frames_bit_exact():
    frames, _, _ = synth.render_orbit_frames(3, 640, 480, arc_deg=6.0)
    _cmp_orb(frames, 1000)


def test_orb_odd_size_and_few_features_bit_exact():
    frames, _, _ = synth.render_orbit_frames(2, 652, 364, arc_deg=2.0, seed=11)   # width % 4 == 0, small levels
    _cmp_orb(frames, 300)


def test_orb_1080p_bit_exact():
    frames, _, _ = synth.render_orbit_frames(1, 1920, 1080, arc_deg=1.0, seed=5, tex_size=2048)
    _cmp_orb(frames, 4000)


def test_orb_4k_8000_keypoints_bit_exact():
    """BASELINE config 5 shape: 3840 x 2160 frames, 8000 key points (the 12-bit x / y packing of the candidate keys and
    the per-level capacities at their largest)."""
    frames, _, _ = synth.render_orbit_frames(2, 3840, 2160, arc_deg=0.5, seed=9, tex_size=4096)
    _cmp_orb(frames, 8000)


def _orb_pyramid_levels(frames, nfeatures, scale_factor=1.2):
    """The pyramid mm_orb_detect_compute leaves at the head of its workspace: list over levels >= 1 of [B, h, pitch] u8."""
    B, H, W = frames.shape
    prm = ops.orb_params(nfeatures)
    prm.scale_factor = scale_factor
    wsp = ops.OrbWorkspace(B, H, W, prm, DEV, brief_pattern())
    wsp.ws.fill_(0xA5)
    ops.orb_detect_compute(dev(frames), wsp)
    torch.cuda.synchronize()
    w, h, _, _ = ops.orb_level_sizes(H, W, prm)
    ws, off, out = wsp.ws.cpu().numpy(), 0, []
    for l in range(1, len(w)):
        pitch = (int(w[l]) + 63) // 64 * 64
        per = (pitch * int(h[l]) + 64 + 255) // 256 * 256
        out.append(np.stack([ws[off + b * per: off + b * per + pitch * int(h[l])].reshape(int(h[l]), pitch) for b in range(B)]))
        off += per * B
    return out, w, h


@pytest.mark.parametrize("W,H,scale", [(640, 480, 1.2), (652, 364, 1.2), (1920, 1080, 1.2), (332, 258, 1.2), (1284, 722, 1.5),
                                       (800, 600, 1.05), (3840, 2160, 1.2)])
def test_orb_pyramid_fused_launches_equal_oracle_and_per_level(W, H, scale, monkeypatch):
    """The pyramid from the fused launches (runs of levels through LDS, halos recomputed) == the per-level kernel == the
    oracle's resize chain, byte for byte, zero row padding included; odd sizes, other scale factors (1.05: the fused
    rectangles do not fit and the per-level kernel is taken), 4K."""
    rng = np.random.default_rng(W + H)
    frames = rng.integers(0, 256, (2, H, W), dtype=np.uint8)
    fused, w, h = _orb_pyramid_levels(frames, 500, scale)
    monkeypatch.setenv("MM_ORB_PYRAMID", "levels")
    per_level, _, _ = _orb_pyramid_levels(frames, 500, scale)
    for l, (a, b) in enumerate(zip(fused, per_level), start=1):
        np.testing.assert_array_equal(a, b, err_msg=f"level {l}")
        assert not a[:, :, int(w[l]):].any()
    if scale == 1.2 and W <= 1920:
        for b in range(2):
            prev = frames[b]
            for l in range(1, len(w)):
                prev = oo.resize(prev, int(w[l]), int(h[l]))
                np.testing.assert_array_equal(fused[l - 1][b][:, :int(w[l])], prev, err_msg=f"frame {b} level {l}")


def test_orb_flat_image_gives_no_keypoints():
    frames = np.full((1, 480, 640), 90, np.uint8)
    n = _orb_gpu(frames, 500)[5]
    assert n[0] == 0


def test_orb_pyramid_matches_oracle_resize():
    """Level geometry from the C ABI equals the oracle's independent computation."""
    prm = ops.orb_params(4000)
    w, h, nf, sc = ops.orb_level_sizes(1080, 1920, prm)
    wo, ho, no, so = oo.level_sizes(1080, 1920, 4000)
    np.testing.assert_array_equal(w, wo)
    np.testing.assert_array_equal(h, ho)
    np.testing.assert_array_equal(nf, no)
    np.testing.assert_array_equal(sc, so)


# ============================================================================================== triangulation

def test_triangulate_dlt_vs_oracle():
    pr = synth.make_ba_problem(30, 500, 6, seed=8)
    F = 30
    proj = np.einsum("ij,fjk->fik", pr["K"], pr["ext"])
    fi = pr["fi"].reshape(500, 6)
    obs = pr["obs"].reshape(500, 6, 2)
    f0, f1 = fi[:, 0].astype(np.int32), fi[:, -1].astype(np.int32)
    x0, x1 = obs[:, 0], obs[:, -1]
    X = ops.triangulate_dlt(dev(proj), dev(f0), dev(f1), dev(x0), dev(x1)).cpu().numpy()
    Xo = bo.triangulate_dlt(proj[f0], proj[f1], x0, x1)
    # f64 Jacobi SVD vs LAPACK SVD of the same 4x4 systems: 1e-8 relative (north star asks 1e-4)
    np.testing.assert_allclose(X, Xo, rtol=1e-8, atol=1e-9)
    assert np.median(np.abs(X - pr["pts_gt"])) < 0.05   # and they are the (noisy-pixel) scene points


# ============================================================================================== BA sweeps

def _problem(F, P, L, seed):
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    cams = bo.frame_parameters(pr["ext"]).reshape(F, 6)
    ctx = default_context()
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV, ctx)
    return pr, cams, pb


def test_ba_residual_matches_golden_and_oracle(golden_dir):
    d = np.load(os.path.join(golden_dir, "g3_point_pose_fun.npz"))
    for tag in ("small", "mid"):
        F, P, L, seed = (int(d[f"{tag}_{k}"]) for k in ("F", "P", "L", "seed"))
        pr, cams, pb = _problem(F, P, L, seed)
        c2, res = pb.residual(dev(cams), dev(pr["pts0"]), True)
        res = res.cpu().numpy()
        # reference pointFun output captured by import: 1e-10 relative
        np.testing.assert_allclose(res[d[f"{tag}_sel"]], d[f"{tag}_res"], rtol=1e-10, atol=1e-9)
        assert abs(0.5 * float(c2) - float(d[f"{tag}_cost"])) <= 1e-11 * float(d[f"{tag}_cost"])
        x = np.hstack([cams.ravel(), pr["pts0"].ravel()])
        np.testing.assert_allclose(res.ravel(), bo.point_fun(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"]),
                                   rtol=1e-10, atol=1e-9)


def test_ba_residual_zero_and_tiny_rotation():
    F, P = 4, 10
    pr = synth.make_ba_problem(F, P, 4, seed=4)
    cams = bo.frame_parameters(pr["ext"]).reshape(F, 6)
    cams[0, :3] = 0.0                 # theta == 0 : nan_to_num branch of the reference (bundleAdjuster.py:19-21)
    cams[1, :3] = [1e-9, 0, 0]
    cams[2, :3] = [3e-3, -2e-3, 1e-3]  # series / closed-form boundary region
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV)
    _, res = pb.residual(dev(cams), dev(pr["pts0"]), True)
    x = np.hstack([cams.ravel(), pr["pts0"].ravel()])
    np.testing.assert_allclose(res.cpu().numpy().ravel(), bo.point_fun(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"]),
                               rtol=1e-10, atol=1e-8)
    Jc, Jp = pb.jacobian(dev(cams), dev(pr["pts0"]))
    Jco, Jpo = bo.jacobian_fd(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"])
    np.testing.assert_allclose(Jc.cpu().numpy(), Jco, rtol=2e-6, atol=2e-4)
    np.testing.assert_allclose(Jp.cpu().numpy(), Jpo, rtol=2e-6, atol=2e-4)


def test_ba_jacobian_vs_central_differences():
    pr, cams, pb = _problem(12, 300, 5, 5)
    x = np.hstack([cams.ravel(), pr["pts0"].ravel()])
    Jc, Jp = pb.jacobian(dev(cams), dev(pr["pts0"]))
    Jco, Jpo = bo.jacobian_fd(x, pr["K"], 12, 300, pr["fi"], pr["pi"], pr["obs"])
    # central differences with h=1e-6 on pixel-scale derivatives (up to ~1e4): truncation+rounding ~1e-4 absolute
    np.testing.assert_allclose(Jc.cpu().numpy(), Jco, rtol=2e-6, atol=5e-4)
    np.testing.assert_allclose(Jp.cpu().numpy(), Jpo, rtol=2e-6, atol=5e-4)


def _dense_normal(pr, F, P, Jc, Jp, res):
    fi, pi = pr["fi"], pr["pi"]
    B = np.zeros((F, 6, 6))
    gc = np.zeros((F, 6))
    C = np.zeros((P, 3, 3))
    gp = np.zeros((P, 3))
    np.add.at(B, fi, np.einsum("omi,omj->oij", Jc, Jc))
    np.add.at(gc, fi, np.einsum("omi,om->oi", Jc, res))
    np.add.at(C, pi, np.einsum("omi,omj->oij", Jp, Jp))
    np.add.at(gp, pi, np.einsum("omi,om->oi", Jp, res))
    return B, gc, C, gp


def test_ba_normal_equations_jvp_schur_backsub_consistent():
    F, P, L = 12, 300, 5
    pr, cams, pb = _problem(F, P, L, 5)
    cd, pd = dev(cams), dev(pr["pts0"])
    Jc, Jp = (a.cpu().numpy() for a in pb.jacobian(cd, pd))
    res = pb.residual(cd, pd, True)[1].cpu().numpy()
    Bo, gco, Co, gpo = _dense_normal(pr, F, P, Jc, Jp, res)
    B, gc, C, gp = pb.normal_eq(cd, pd)
    np.testing.assert_allclose(B.cpu().numpy(), Bo, rtol=1e-11, atol=1e-6)
    np.testing.assert_allclose(gc.cpu().numpy(), gco, rtol=1e-11, atol=1e-6)
    tri = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    np.testing.assert_allclose(C.cpu().numpy(), np.stack([Co[:, i, j] for i, j in tri], 1), rtol=1e-11, atol=1e-6)
    np.testing.assert_allclose(gp.cpu().numpy(), gpo, rtol=1e-11, atol=1e-6)
    # J w
    rng = np.random.default_rng(0)
    wc, wp = rng.normal(size=(F, 6)), rng.normal(size=(P, 3))
    jv = pb.jvp(cd, pd, dev(wc), dev(wp)).cpu().numpy()
    jvo = np.einsum("omi,oi->om", Jc, wc[pr["fi"]]) + np.einsum("omi,oi->om", Jp, wp[pr["pi"]])
    np.testing.assert_allclose(jv, jvo, rtol=1e-11, atol=1e-8)
    # damped system solved through Schur + Cholesky + back-substitution == dense solve of the full normal equations
    reg = 1e-3
    n = 6 * F + 3 * P
    H = np.zeros((n, n))
    for f in range(F):
        H[6 * f:6 * f + 6, 6 * f:6 * f + 6] = Bo[f]
    for p in range(P):
        H[6 * F + 3 * p:6 * F + 3 * p + 3, 6 * F + 3 * p:6 * F + 3 * p + 3] = Co[p]
    for o, (f, p) in enumerate(zip(pr["fi"], pr["pi"])):
        E = Jc[o].T @ Jp[o]
        H[6 * f:6 * f + 6, 6 * F + 3 * p:6 * F + 3 * p + 3] += E
        H[6 * F + 3 * p:6 * F + 3 * p + 3, 6 * f:6 * f + 6] += E.T
    Dm = np.concatenate([np.sqrt(np.einsum("fii->fi", Bo)).ravel(), np.sqrt(np.einsum("pii->pi", Co)).ravel()])
    Hd = H + reg * np.diag(Dm ** 2)
    g = np.concatenate([gco.ravel(), gpo.ravel()])
    sol = np.linalg.solve(Hd, g)
    Bd = B.clone()
    Bd.diagonal(dim1=1, dim2=2).add_(reg * torch.diagonal(B, dim1=1, dim2=2))
    Cd = C.clone()
    Cd[:, [0, 3, 5]] += reg * C[:, [0, 3, 5]]
    S, v, Cinv = pb.schur(cd, pd, Bd, Cd, gc, gp)
    Sl = np.tril(S.cpu().numpy())
    Sfull = Sl + np.tril(Sl, -1).T
    Hcc, Hcp, Hpp = Hd[:6 * F, :6 * F], Hd[:6 * F, 6 * F:], Hd[6 * F:, 6 * F:]
    So = Hcc - Hcp @ np.linalg.solve(Hpp, Hcp.T)
    np.testing.assert_allclose(Sfull, So, rtol=1e-9, atol=1e-6 * np.abs(So).max())
    # the general (LDS-atomic) kernel fills all of S and agrees with the pair-list kernel
    pb2 = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV, pairs=False)
    assert pb.n_pairs > 0 and pb2.n_pairs == 0
    S2, v2, _ = pb2.schur(cd, pd, Bd, Cd, gc, gp)
    np.testing.assert_allclose(S2.cpu().numpy(), So, rtol=1e-9, atol=1e-6 * np.abs(So).max())
    np.testing.assert_allclose(v2.cpu().numpy(), v.cpu().numpy(), rtol=1e-10, atol=1e-9 * float(v.abs().max()))
    S_again, _, _ = pb.schur(cd, pd, Bd, Cd, gc, gp)
    assert torch.equal(S_again, S)          # the pair-list kernel is bitwise reproducible
    info = ops.chol_solve(S, v, half_bandwidth=6 * pb.cam_span + 5)
    assert int(info) == 0
    dc = v.reshape(F, 6)
    dp = pb.backsub(cd, pd, Cinv, gp, dc)
    got = np.concatenate([dc.cpu().numpy().ravel(), dp.cpu().numpy().ravel()])
    np.testing.assert_allclose(got, sol, rtol=1e-6, atol=1e-9 * np.abs(sol).max())
    # the back-substitution over observations (workspace given: what pb.backsub calls) and the one-launch form with a
    # thread per point (ws = NULL) are the same sums in the same order
    from meatmodeler_amd._lib import lib, ptr
    import ctypes as C
    dp1 = torch.empty_like(dp)
    pb.ctx.check(lib.mm_ba_backsub(pb.ctx.h, C.byref(pb.pb), ptr(cd), ptr(pd), ptr(Cinv), ptr(gp), ptr(dc), ptr(dp1), None, 0),
                 "mm_ba_backsub")
    np.testing.assert_allclose(dp1.cpu().numpy(), dp.cpu().numpy(), rtol=1e-12, atol=1e-14 * float(dp.abs().max()))
    assert torch.equal(pb.backsub(cd, pd, Cinv, gp, dc), dp)          # reproducible


def test_ba_full_size_adjoint_and_reduced_system_properties():
    """BASELINE size (500 cameras, 300 k points, 1.5 M observations): properties that need no oracle run --
    the sweeps are mutually adjoint (<J w, r> == <w, J^T r> with J^T r from the normal-equation sweep, J w from the
    JVP sweep), the cost is the squared norm of the residual vector, and the overlapped build + banded solve returns
    the solution of the reduced camera system it built (|S dc - v| small against |v|)."""
    F, P, L = 500, 300_000, 5
    pr = synth.make_ba_problem(F, P, L, seed=9)
    cams = dev(bo.frame_parameters(pr["ext"]).reshape(F, 6))
    pts = dev(pr["pts0"])
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV)
    cost2, res = pb.residual(cams, pts, True)
    assert abs(float(cost2) - float((res * res).sum())) <= 1e-11 * float(cost2)
    B, gc, C6, gp = pb.normal_eq(cams, pts)
    rng = np.random.default_rng(1)
    wc, wp = dev(rng.normal(size=(F, 6))), dev(rng.normal(size=(P, 3)))
    Jw = pb.jvp(cams, pts, wc, wp)
    lhs = float((Jw * res).sum())
    rhs = float((wc * gc).sum() + (wp * gp).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.0)
    Bd = B + 1e-4 * torch.diag_embed(torch.diagonal(B, dim1=1, dim2=2))
    Cd = C6.clone()
    Cd[:, [0, 3, 5]] *= 1.0 + 1e-4
    hb = 6 * pb.cam_span + 5
    S, v, _ = pb.schur(cams, pts, Bd, Cd, gc, gp)
    Sl = torch.tril(S)
    Sfull = Sl + torch.tril(S, -1).T
    v0 = v.clone()
    info, dc, _ = pb.schur_solve(cams, pts, Bd, Cd, gc, gp, hb)
    assert int(info) == 0
    r = Sfull @ dc - v0
    assert float(r.norm()) <= 1e-9 * float(v0.norm())


def test_device_built_indices_equal_host_builders():
    """CSR by point / camera and the co-observation pair list built on the device (torch sorts + two HIP kernels)
    are identical to the host C++ builders (mm_ba_build_index / mm_ba_build_pairs)."""
    import ctypes as C
    from meatmodeler_amd._lib import lib, c_i32p, c_i64p
    rng = np.random.default_rng(9)
    for point_major in (True, False):
        pr = synth.make_ba_problem(15, 200, 5, seed=3)
        fi, pi, obs = pr["fi"].astype(np.int32), pr["pi"].astype(np.int32), pr["obs"]
        if not point_major:
            perm = rng.permutation(len(fi))
            fi, pi, obs = fi[perm], pi[perm], obs[perm]
        pb = ops.BADevice(pr["K"], fi, pi, obs, 15, 200, DEV)
        pt_ptr, pt_obs, cam_ptr, cam_obs = ops.ba_build_index(15, 200, fi, pi)
        np.testing.assert_array_equal(pb.pt_ptr.cpu().numpy(), pt_ptr)
        np.testing.assert_array_equal(pb.pt_obs.cpu().numpy(), pt_obs)
        np.testing.assert_array_equal(pb.cam_ptr.cpu().numpy(), cam_ptr)
        np.testing.assert_array_equal(pb.cam_obs.cpu().numpy(), cam_obs)
        span = pb.cam_span
        seg_ptr = np.zeros(15 * (span + 1) + 1, np.int64)
        args = (15, 200, len(fi), fi.ctypes.data_as(c_i32p), pi.ctypes.data_as(c_i32p), pt_ptr.ctypes.data_as(c_i32p),
                np.ascontiguousarray(pt_obs).ctypes.data_as(c_i32p), cam_ptr.ctypes.data_as(c_i32p),
                np.ascontiguousarray(cam_obs).ctypes.data_as(c_i32p), span, seg_ptr.ctypes.data_as(c_i64p))
        n = lib.mm_ba_build_pairs(*args, None, None, 0)
        assert n == pb.n_pairs and n > 0
        po, po2 = np.zeros(n, np.int32), np.zeros(n, np.int32)
        lib.mm_ba_build_pairs(*args, po.ctypes.data_as(c_i32p), po2.ctypes.data_as(c_i32p), n)
        np.testing.assert_array_equal(pb.pair_o.cpu().numpy(), po)
        np.testing.assert_array_equal(pb.pair_o2.cpu().numpy(), po2)
        ne = np.flatnonzero(np.diff(seg_ptr) > 0)
        np.testing.assert_array_equal(pb.seg_ids.cpu().numpy(), ne)
        cb, ce, cs = pb.chunk_begin.cpu().numpy(), pb.chunk_end.cpu().numpy(), pb.chunk_seg.cpu().numpy()
        assert (ce - cb).max() <= 256 and (ce - cb).min() >= 1 and (ce - cb).sum() == n
        np.testing.assert_array_equal(cb[pb.seg_chunk_ptr.cpu().numpy()[:-1]], seg_ptr[ne])
        assert (np.diff(cs) >= 0).all()
    with pytest.raises(ValueError):
        ops.BADevice(pr["K"], np.array([15], np.int32), np.array([0], np.int32), np.zeros((1, 2)), 15, 200, DEV)


@pytest.mark.parametrize("n", [2, 64, 130, 500, 1000])
def test_chol_solve_random_spd(n):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    Ad, bd = dev(A), dev(b)
    info = ops.chol_solve(Ad, bd)
    assert int(info) == 0
    np.testing.assert_allclose(bd.cpu().numpy(), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    L = np.tril(Ad.cpu().numpy())
    np.testing.assert_allclose(L @ L.T, A, rtol=1e-10, atol=1e-9 * n)


@pytest.mark.parametrize("n,hb", [(500, 40), (1000, 130), (770, 63), (320, 5), (3000, 528), (1500, 900), (2000, 1100)])
def test_chol_solve_banded(n, hb):
    """Bands up to 15 blocks of 64 take the single-launch data-flow factorisation (3000/528 is the shape of the
    500-frame clip), wider ones the launch-per-column path."""
    rng = np.random.default_rng(n + hb)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))     # banded factor -> banded SPD product
    A = M @ M.T + n * np.eye(n)
    i, j = np.indices((n, n))
    assert (A[np.abs(i - j) > hb] == 0).all()
    b = rng.normal(size=n)
    Ad, bd = dev(A), dev(b)
    info = ops.chol_solve(Ad, bd, half_bandwidth=hb)
    assert int(info) == 0
    np.testing.assert_allclose(bd.cpu().numpy(), np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    L = np.tril(Ad.cpu().numpy())
    L[i - j > hb] = 0.0
    np.testing.assert_allclose(L, np.linalg.cholesky(A), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("n,hb", [(3000, 528), (770, 63), (320, 5)])
def test_chol_launch_per_column_path_agrees_with_single_launch(n, hb):
    """The narrow-band shapes again with the context told to avoid the single-launch factorisation (whose cross-workgroup
    hand-over leans on gfx942 / gfx950 memory-system behaviour, see the guard at the top of chol.hip): the launch-per-column
    path -- ordinary kernel boundaries, no hand-over inside a launch -- gives the same solution to rounding, so a toolchain or
    ASIC change that broke the hand-over would show up as a difference between the two."""
    _needs_single_launch_chol()
    from meatmodeler_amd._lib import default_context
    ctx = default_context()
    rng = np.random.default_rng(n + hb)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    sols = []
    for avoid in (0, 1):
        ctx.control(ctx.CTL_CHOL_AVOID_FUSED, avoid)
        try:
            Ad, bd = dev(A), dev(b)
            assert int(ops.chol_solve(Ad, bd, half_bandwidth=hb)) == 0
            ctx.sync()
            sols.append((bd.cpu().numpy(), int(ctx.control(ctx.CTL_CHOL_LAST_PATH))))
        finally:
            ctx.control(ctx.CTL_CHOL_AVOID_FUSED, 0)
    assert sols[0][1] != sols[1][1], "the two runs must have taken different factorisation paths"
    np.testing.assert_allclose(sols[0][0], sols[1][0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(sols[1][0], np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("both", [True, False])
@pytest.mark.parametrize("n,hb", [(3000, 528), (1000, 130), (770, 63), (320, 5), (500, 40), (1200, 300), (2048, 64),
                                   (1500, 900)])
def test_chol_solve_sym_two_ended(n, hb, both):
    """mm_chol_solve_sym (solution only): narrow bands are eliminated from both ends of the matrix at once (two sets of
    workgroups, the second in reversed coordinates in the upper triangle; n that is not a multiple of 64 is padded
    virtually).  3000/528 is the reduced camera system of the 500-frame clip; 1500/900 is too wide for the
    single-launch path and takes the launch-per-column factorisation.  `both` = False: only the lower triangle is
    given (the upper one holds garbage that must be ignored / overwritten)."""
    rng = np.random.default_rng(n + hb)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    Ain = A if both else np.tril(A) + np.triu(rng.normal(size=(n, n)), 1)
    ref = np.linalg.solve(A, b)
    for rep in range(2):
        Ad, bd = dev(Ain), dev(b)
        info = ops.chol_solve_sym(Ad, bd, half_bandwidth=hb, both_triangles=both)
        assert int(info) == 0
        np.testing.assert_allclose(bd.cpu().numpy(), ref, rtol=1e-9, atol=1e-12)


def test_chol_solve_sym_reports_non_spd():
    n, hb = 1000, 130
    rng = np.random.default_rng(5)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))
    A = M @ M.T + n * np.eye(n)
    for bad_col in (100, 930, 500):            # eliminated by side 0, by side 1 (reversed), in the middle block
        Ab = A.copy()
        Ab[bad_col, bad_col] = -1.0
        info = ops.chol_solve_sym(dev(Ab), dev(np.ones(n)), half_bandwidth=hb, both_triangles=True)
        assert int(info) > 0, (bad_col, int(info))


@pytest.mark.parametrize("F,P,L", [(200, 6000, 12), (130, 2000, 40)])
def test_schur_solve_overlapped_equals_sequential(F, P, L):
    """mm_ba_schur_solve (S built in camera slabs on one stream while the single-launch Cholesky consumes finished
    block rows on another) returns bit for bit what mm_ba_schur followed by mm_chol_solve returns."""
    pr = synth.make_ba_problem(F, P, L, seed=F)
    cams = dev(bo.frame_parameters(pr["ext"]).reshape(F, 6))
    pts = dev(pr["pts0"])
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV)
    assert pb.slabs is not None and pb.n_pairs > 0
    pb.overlap = True
    B, gc, C6, gp = pb.normal_eq(cams, pts)
    Bd = B + 1e-3 * torch.diag_embed(torch.diagonal(B, dim1=1, dim2=2))
    Cd = C6.clone()
    Cd[:, [0, 3, 5]] *= 1.0 + 1e-3
    hb = 6 * pb.cam_span + 5
    S, v, Cinv0 = pb.schur(cams, pts, Bd, Cd, gc, gp)
    Sh = S.cpu().numpy()
    band = np.abs(np.subtract.outer(np.arange(6 * F), np.arange(6 * F))) <= hb
    # both triangles of the band are produced: off-diagonal camera blocks as exact mirror images, the diagonal 6 x 6
    # blocks (sum of Y E^T products, rounded entry by entry) symmetric to rounding
    assert not Sh[~band].any()
    blockdiag = np.kron(np.eye(F), np.ones((6, 6))).astype(bool)
    assert np.array_equal(Sh[~blockdiag], Sh.T[~blockdiag])
    np.testing.assert_allclose(Sh, Sh.T, rtol=1e-11, atol=1e-12 * np.abs(Sh).max())
    info0 = ops.chol_solve_sym(S, v, half_bandwidth=hb, both_triangles=True)
    ref = v.clone()
    np.testing.assert_allclose(ref.cpu().numpy(), np.linalg.solve(Sh, pb.schur(cams, pts, Bd, Cd, gc, gp)[1].cpu().numpy()),
                               rtol=1e-7, atol=1e-12)
    for _ in range(3):      # repeated: the flags / workspaces are reused
        info, dc, Cinv = pb.schur_solve(cams, pts, Bd, Cd, gc, gp, hb)
        assert int(info) == 0 and int(info0) == 0
        assert torch.equal(dc, ref) and torch.equal(Cinv, Cinv0)


def test_schur_solve_falls_back_when_kernels_are_serialised():
    """The overlapped build + solve needs two kernels in flight at once.  With kernel launches serialised (here by the
    runtime's AMD_SERIALIZE_KERNEL debug switch; rocprofv3 --pmc does the same) the consumer's bounded spins give up,
    the driver notices (info = -1), warns and continues with the two steps one after the other."""
    _needs_single_launch_chol()
    import subprocess
    import sys
    code = (
        "import warnings, numpy as np, torch\n"
        "from meatmodeler_amd import ops, synth\n"
        "from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters\n"
        "pr = synth.make_ba_problem(130, 2000, 12, seed=3)\n"
        "dev = torch.device('cuda:0')\n"
        "pb = ops.BADevice(pr['K'], pr['fi'], pr['pi'], pr['obs'], 130, 2000, dev)\n"
        "pb.overlap = True\n"
        "assert pb.slabs is not None\n"
        "cams = torch.as_tensor(frameParameters(pr['ext']).reshape(130, 6)).to(dev)\n"
        "pts = torch.as_tensor(pr['pts0']).to(dev)\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    res = SchurTRF(pb).solve(cams, pts, max_nfev=4)\n"
        "assert any('serialised' in str(x.message) for x in w), [str(x.message) for x in w]\n"
        "assert not pb.overlap and np.isfinite(res.cost) and res.nfev == 4\n"
        "print('fallback ok', res.cost)\n")
    env = dict(os.environ, AMD_SERIALIZE_KERNEL="3")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n,k,split", [(1, 1, 0), (1000, 3, 400), (1_234_567, 5, 3000), (300_001, 8, 300_001), (50_000, 11, 7)])
def test_multi_dot_vs_numpy(n, k, split):
    """mm_multi_dot: k inner products in one launch, reported as (i < split, i >= split, total); more than 8 pairs are
    split over several launches by the wrapper; repeated calls reuse the workspace (its counter resets itself)."""
    rng = np.random.default_rng(n + k)
    A = rng.normal(size=(k, n))
    B = rng.normal(size=(k, n))
    md = ops.MultiDot(DEV)
    pairs = [(dev(A[i]), dev(B[i])) for i in range(k)]
    for _ in range(2):
        out = md(pairs, split).cpu().numpy()
        want = np.array([[A[i, :split] @ B[i, :split], A[i, split:] @ B[i, split:], A[i] @ B[i]] for i in range(k)])
        np.testing.assert_allclose(out, want, rtol=1e-11, atol=1e-9)
    out2 = md(pairs, split).cpu().numpy()
    assert np.array_equal(out, out2)            # deterministic


def test_trf_fused_passes_vs_torch():
    """mm_trf_fused: the element-wise passes of the 2-D subspace step and the inner products they carry, against the
    same formulas written with torch (outputs bit for bit: IEEE division / sqrt; sums to rounding)."""
    rng = np.random.default_rng(3)
    n, split = 200_003, 3000
    pr = synth.make_ba_problem(6, 40, 4, seed=1)
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], 6, 40, DEV)
    g, si, x = dev(rng.normal(size=n)), dev(rng.uniform(0.5, 2.0, size=n)), dev(rng.normal(size=n))
    gh, ghs = torch.empty_like(g), torch.empty_like(g)
    r0 = pb.trf_fused(0, [g, si], [gh, ghs], split=split)
    assert torch.equal(gh, g / si) and torch.equal(ghs, (g / si) / si)
    np.testing.assert_allclose(r0[0].cpu().numpy(), [float(gh[:split] @ gh[:split]), float(gh[split:] @ gh[split:]), float(gh @ gh)], rtol=1e-12)
    assert r0[1].cpu().tolist() == [float(g[:split].abs().max()), float(g[split:].abs().max()), float(g.abs().max())]
    gh2 = r0[0, 2:3].contiguous()
    v, dp = dev(rng.normal(size=split)), dev(rng.normal(size=n - split))
    gn, q1, w = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    r1 = pb.trf_fused(1, [v, dp, si, gh], [gn, q1], [gh2], split=split)
    assert torch.equal(gn, torch.cat([v, dp]) * si) and torch.equal(q1, gh / torch.sqrt(gh2))
    np.testing.assert_allclose(r1[:2, 2].cpu().numpy(), [float(q1 @ gn), float(gn @ gn)], rtol=1e-11)
    sc = r1[0, 2:3].contiguous()
    r2 = pb.trf_fused(2, [gn, q1], [w], [sc], split=split)
    np.testing.assert_allclose(w.cpu().numpy(), (gn - sc * q1).cpu().numpy(), rtol=0, atol=1e-15)   # (one FMA vs mul + sub)
    np.testing.assert_allclose(float(r2[0, 2]), float(w @ w), rtol=1e-11)
    wn2 = r2[0, 2:3].contiguous()
    q2, s1, s2 = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    r3 = pb.trf_fused(3, [w, q1, si, gh, x], [q2, s1, s2], [wn2], split=split)
    assert torch.equal(q2, w / torch.sqrt(wn2)) and torch.equal(s1, q1 / si) and torch.equal(s2, q2 / si)
    np.testing.assert_allclose(r3[:5, 2].cpu().numpy(), [float(s1 @ s1), float(s1 @ s2), float(s2 @ s2), float(q2 @ gh), float(x @ x)],
                               rtol=1e-10, atol=1e-12)
    xn = torch.empty_like(x)
    pb.trf_fused(4, [x, s1, s2], [xn], h0=0.3, h1=-1.7, split=split)
    np.testing.assert_allclose(xn.cpu().numpy(), (x + 0.3 * s1 - 1.7 * s2).cpu().numpy(), rtol=1e-15, atol=1e-15)


def test_profile_levels_select_what_is_bracketed():
    """mm_profile_enable: 1 = every launch, 3 = only the kernel named by mm_profile_select (what bench.py's timed steps
    use: one kernel's events instead of a bubble behind every launch), 0 = nothing."""
    ctx = default_context()
    rng = np.random.default_rng(5)
    n = 100_000
    a, b = dev(rng.normal(size=n)), dev(rng.uniform(0.5, 2.0, size=n))
    pr = synth.make_ba_problem(4, 10, 3, seed=1)
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], 4, 10, DEV)
    md = ops.MultiDot(DEV)

    def work():
        pb.trf_fused(0, [a, b], [torch.empty_like(a), torch.empty_like(a)], split=600)
        md([(a, b)], 600)

    try:
        ctx.profile(1)
        work()
        every = ctx.profile_report()
        assert every["fused_vec_kernel"][0] == 1 and every["multi_dot_kernel"][0] == 1
        assert every["fused_vec_kernel"][1] > 0
        ctx.profile(3, only="multi_dot_kernel")
        work()
        work()
        only = ctx.profile_report()
        assert list(only) == ["multi_dot_kernel"] and only["multi_dot_kernel"][0] == 2
        ctx.profile(0)
        work()
        assert ctx.profile_report() == only          # nothing new is recorded while profiling is off
    finally:
        ctx.profile(0)


def test_vector_kernels_on_empty_vectors():
    md = ops.MultiDot(DEV)
    e = torch.empty(0, dtype=torch.float64, device=DEV)
    assert md([(e, e), (e, e)], 0).cpu().tolist() == [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    pr = synth.make_ba_problem(4, 10, 3, seed=1)
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], 4, 10, DEV)
    r = pb.trf_fused(0, [e, e], [e.clone(), e.clone()], split=0)
    assert r.cpu().tolist() == [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]


def test_jvp_dots_equals_jvp_plus_inner_products():
    pr = synth.make_ba_problem(30, 3000, 5, seed=12)
    F, P = 30, 3000
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], F, P, DEV)
    rng = np.random.default_rng(3)
    cams, pts = dev(bo.frame_parameters(pr["ext"]).reshape(F, 6)), dev(pr["pts0"])
    wc, wp = dev(rng.normal(size=(F, 6))), dev(rng.normal(size=(P, 3)))
    ref = pb.jvp(cams, pts, wc, wp)
    other = dev(rng.normal(size=(pb.O, 2)))
    for rep in range(2):                     # (the workspace counter resets itself)
        out, rows = pb.jvp_dots(cams, pts, wc, wp)
        assert torch.equal(out, ref)
        r = rows.cpu().numpy()
        np.testing.assert_allclose(r[:, 2], [float((ref * ref).sum())] * 2, rtol=1e-13)
        assert (r[:, 0] == 0).all() and (r[:, 1] == r[:, 2]).all()
        out, rows = pb.jvp_dots(cams, pts, wc, wp, other=other)
        np.testing.assert_allclose(rows.cpu().numpy()[:, 2], [float((ref * other).sum()), float((ref * ref).sum())], rtol=1e-12)


def test_trf_damping_vs_scipy_formula():
    """mm_trf_damping == the scalar recipe of SciPy trf.py:473-477 (regulariser from the Cauchy-like model along g_h)."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        gh2, d11 = float(rng.uniform(1e-3, 1e6)), float(rng.uniform(0, 1e8)) * float(rng.integers(0, 2) + rng.integers(0, 2))
        Delta, floor = float(10 ** rng.uniform(-3, 4)), 1e-9
        a, b = 0.5 * d11, -gh2
        to_tr = Delta / np.sqrt(gh2)
        ts = [0.0, to_tr]
        if a != 0 and 0.0 < -0.5 * b / a < to_tr:
            ts.append(-0.5 * b / a)
        reg = -min(t * (a * t + b) for t in ts) / Delta ** 2
        out = ops.trf_damping(dev(np.array([gh2])), dev(np.array([d11])), Delta, floor).cpu().numpy()
        np.testing.assert_allclose(out, [reg, max(reg, floor)], rtol=1e-13, atol=0)


def test_band_view_addresses_lower_band():
    pr = synth.make_ba_problem(12, 80, 4, seed=2)
    pb = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], 12, 80, DEV)
    n = 72
    pb._alloc_S(n)
    S = torch.arange(n * n, dtype=torch.float64, device=DEV).view(n, n)
    pb._S.copy_(S)
    hb = 17
    band = pb.band_view(hb).cpu().numpy()
    Sh = S.cpu().numpy()
    for j in (0, 5, 40, 71):
        for k in (0, 1, hb):
            assert band[j, k] == (Sh[j + k, j] if j + k < n else 0.0)


@pytest.mark.parametrize("n,hb", [(700, 130), (900, 2000)])
def test_chol_solve_several_right_hand_sides(n, hb):
    """nrhs > 1: the first right-hand side rides along in the single-launch factorisation, the others take the
    launch-per-column forward substitution; all take the single-launch backward substitution when the band allows."""
    rng = np.random.default_rng(n)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -min(hb, n - 1) // 2))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=(3, n))
    Ad, bd = dev(A), dev(b)
    info = ops.chol_solve(Ad, bd, half_bandwidth=min(hb, n))
    assert int(info) == 0
    np.testing.assert_allclose(bd.cpu().numpy(), np.linalg.solve(A, b.T).T, rtol=1e-9, atol=1e-12)


def test_chol_reports_non_spd():
    A = np.eye(70)
    A[66, 66] = -1.0
    info = ops.chol_solve(dev(A), dev(np.ones(70)))
    assert int(info) == 67


# ============================================================================================== adjustPoints

def _similarity_align(X, Y):
    """Least-squares similarity (7-DoF gauge) mapping X onto Y."""
    mx, my = X.mean(0), Y.mean(0)
    Xc, Yc = X - mx, Y - my
    U, S, Vt = np.linalg.svd(Yc.T @ Xc)
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    s = np.trace(np.diag(S) @ D) / (Xc ** 2).sum()
    return s * Xc @ R.T + my


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_adjust_points_vs_reference_golden(golden_dir, tag):
    d = np.load(os.path.join(golden_dir, f"g5_adjust_points_{tag}.npz"))
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pts, ext = bundleAdjuster.adjustPoints(pr["ext"], pr["K"], pr["pts0"][:, None, :], pr["obs"], pr["fi"], pr["pi"])
    table = buf.getvalue()
    assert table.splitlines()[0].split() == ["Iteration", "Total", "nfev", "Cost", "Cost", "reduction", "Step", "norm",
                                             "Optimality"]
    assert pts.shape == (P, 3) and len(ext) == F and ext[0].shape == (4, 4)
    # cost / reprojection error at the returned solution, evaluated by the ORACLE cost function
    cams = np.array([np.concatenate([bo.frame_parameters(e[None, :3])[:3], e[:3, 3]]) for e in ext])
    x = np.hstack([cams.ravel(), pts.ravel()])
    cost = 0.5 * np.sum(bo.point_fun(x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"]) ** 2)
    cost_ref = float(d["cost_ref"])
    # north star: reprojection error within 1e-4 relative of the reference's result, and not worse beyond that
    assert cost <= cost_ref * (1 + 1e-4), (cost, cost_ref)
    assert abs(np.sqrt(cost) - np.sqrt(cost_ref)) <= 1e-4 * np.sqrt(cost_ref)
    # 3-D points: the problem has a 7-DoF gauge freedom and the reference's own result is only reproducible to
    # ~1e-3 in x (tests/test_oracle_golden.py::test_g5); compare directly at that level and at 1e-4 after removing
    # the gauge (similarity alignment).
    ref = d["points"]
    scale = np.abs(ref).max()
    # Both runs stop at ftol=1e-4, i.e. NOT at the minimiser: the exact Schur solve moves fully along weakly
    # determined directions that SciPy's truncated LSMR barely touches, so points agree only to ~1e-2 of the scene
    # size here while the cost agrees to 1e-4 (measured: 4e-4 / 3e-3 on cases a / c).
    # The 1e-4 point tolerance of the north star is asserted where it is well posed — both sides at the minimiser —
    # in test_adjust_points_tight_vs_reference_minimiser below; this run only guards against gross errors.
    assert np.abs(pts - ref).max() <= 1e-2 * scale, np.abs(pts - ref).max() / scale
    aligned = _similarity_align(pts, ref)
    assert np.abs(aligned - ref).max() <= 1e-2 * scale, np.abs(aligned - ref).max() / scale
    # ... and it is within the same distance of the reference's converged minimiser as the reference's own early stop
    tight = d["x_tight"][6 * F:].reshape(P, 3)
    d_ref = np.abs(_similarity_align(ref, tight) - tight).max()
    d_own = np.abs(_similarity_align(pts, tight) - tight).max()
    assert d_own <= max(3.0 * d_ref, 1e-4 * scale), (d_own / scale, d_ref / scale)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_adjust_points_tight_vs_reference_minimiser(golden_dir, tag):
    """North star: "3-D points and reprojection error within 1e-4 rel".  Golden G5(ii) is the reference's pointFun +
    sparsity (bundleAdjuster.py:81-102,55-78,180-192) driven by SciPy to the minimiser (status_tight = 2, two
    differently configured runs agree to < 1e-6 modulo gauge; tests/golden/make_golden.py).  The HIP solver is run to
    ftol = xtol = gtol = 1e-12 from the same x0: cost within 1e-8 rel, reprojection RMS within 1e-8 rel, 3-D points
    within 1e-4 rel of the scene size after removing the 7-DoF gauge the reference leaves free (measured ~1e-7)."""
    d = np.load(os.path.join(golden_dir, f"g5_adjust_points_{tag}.npz"))
    assert int(d["status_tight"]) > 0
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], ftol=1e-12,
                                     xtol=1e-12, gtol=1e-12, max_nfev=200, verbose=0)
    ct = float(d["cost_tight"])
    # cost at the returned x evaluated by the ORACLE's cost function (not the solver's own bookkeeping)
    cost = 0.5 * np.sum(bo.point_fun(res.x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"]) ** 2)
    assert abs(cost - ct) <= 1e-8 * ct, (cost, ct)
    assert abs(res.cost - ct) <= 1e-8 * ct
    assert abs(np.sqrt(cost) - np.sqrt(ct)) <= 1e-8 * np.sqrt(ct)
    ref = d["x_tight"][6 * F:].reshape(P, 3)
    scale = np.abs(ref).max()
    pts = res.x[6 * F:].reshape(P, 3)
    err = np.abs(_similarity_align(pts, ref) - ref).max() / scale
    print(f"tight BA {tag}: nfev {res.nfev} status {res.status} cost rel {abs(cost - ct) / ct:.2e} aligned point err {err:.2e}")
    assert err <= 1e-4, err
    assert res.nfev < 200


def test_adjust_points_case_d_two_ended_factorisation_vs_reference(golden_dir):
    """Golden G5 case d (120 cameras, 6000 points, tracks of 8): the reduced camera system has 12 block columns and a
    band of one, so the two-ended factorisation runs inside the solver.  Reference settings: same nfev / status, cost
    within 1e-5 of the reference's run.  Driven to the minimiser: cost within 1e-8 and (on the stored every-10th-point
    subsample, similarity aligned) 3-D points within 1e-4 rel of the reference's converged minimiser."""
    d = np.load(os.path.join(golden_dir, "g5_adjust_points_d.npz"))
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    assert (6 * F + 63) // 64 - 1 >= 4          # nblk - bwb >= 4: mm_chol_solve_sym takes the two-ended path
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], verbose=0)
    assert res.status == int(d["status_ref"]) and res.nfev == int(d["nfev_ref"])
    # (the exact Gauss-Newton steps land closer to the minimum than the reference's truncated LSMR steps in the same
    # number of evaluations: 9675.12032 vs 9675.21915, minimum 9675.12031)
    assert float(d["cost_tight"]) * (1 - 1e-9) <= res.cost <= float(d["cost_ref"]) * (1 + 1e-5)
    assert abs(res.cost - float(d["cost_ref"])) <= 1e-4 * float(d["cost_ref"])
    res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], ftol=1e-12,
                                     xtol=1e-12, gtol=1e-12, max_nfev=200, verbose=0)
    ct = float(d["cost_tight"])
    cost = 0.5 * np.sum(bo.point_fun(res.x, pr["K"], F, P, pr["fi"], pr["pi"], pr["obs"]) ** 2)
    assert abs(cost - ct) <= 1e-8 * ct and abs(res.cost - ct) <= 1e-8 * ct, (cost, res.cost, ct)
    ref = d["points_tight_sub"]
    pts = res.x[6 * F:].reshape(P, 3)[d["sub"]]
    err = np.abs(_similarity_align(pts, ref) - ref).max() / np.abs(ref).max()
    assert err <= 1e-4, err


def test_adjust_points_iterates_follow_scipy(golden_dir):
    """Same trust-region algorithm, exact instead of LSMR inner solve: the iteration table matches the reference's."""
    meta = json.load(open(os.path.join(golden_dir, "g5_adjust_points_meta.json")))
    d = np.load(os.path.join(golden_dir, "g5_adjust_points_c.npz"))
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], verbose=0)
    assert res.status == int(d["status_ref"]) and res.nfev == int(d["nfev_ref"])
    assert abs(res.cost - float(d["cost_ref"])) <= 1e-5 * float(d["cost_ref"])
    ref_costs = [float(line.split()[2]) for line in meta["c"]["table"].splitlines()[1:1 + int(d["nfev_ref"])]]
    assert abs(ref_costs[-1] - res.cost) <= 1e-3 * res.cost


def test_adjust_points_on_real_matches_vs_scipy_recipe(golden_dir):
    """Outlier-laden problem from the oracle front end (C ORB + BF match on a rendered 16-frame clip, no outlier
    rejection — as the reference) solved by the reference's SciPy recipe on the CPU (tools/cpu_reference_ba_probe.py:
    174 function evaluations, 74 s).  The trust-region path is chaotic on such data, so only the outcome is compared:
    the HIP solver must end at a cost no worse than SciPy's within 2 %, in a comparable number of evaluations."""
    d = np.load(os.path.join(golden_dir, "o1_real_match_ba.npz"))
    res = bundleAdjuster.solvePoints(d["ext"], d["K"], d["pts0"], d["obs"], d["fi"], d["pi"], verbose=0)
    print("real-match BA: nfev", res.nfev, "vs", int(d["nfev_ref"]), "cost", res.cost, "vs", float(d["cost_ref"]))
    assert res.status == int(d["status_ref"]) == 2
    assert res.cost <= float(d["cost_ref"]) * 1.02
    assert res.nfev <= 3 * int(d["nfev_ref"])
    res2 = bundleAdjuster.solvePoints(d["ext"], d["K"], d["pts0"], d["obs"], d["fi"], d["pi"], verbose=0)
    assert res2.nfev == res.nfev and res2.cost == res.cost          # bitwise reproducible


def test_damped_step_on_real_matches_equals_oracle_derived_step(golden_dir):
    """The chaos guards above compare outcomes only; this pins the ARITHMETIC of one trust-region step on the same
    outlier-laden data, deterministically: the Jacobian blocks against central differences of the oracle's residual
    function (a sample of observations), then the damped Gauss-Newton step the kernels produce -- normal equations, Schur
    complement, banded Cholesky, back-substitution -- against the same step assembled and solved in NumPy from those blocks
    (point blocks eliminated point by point, dense 96 x 96 reduced system).  A 1 % error anywhere in that chain fails here."""
    d = np.load(os.path.join(golden_dir, "o1_real_match_ba.npz"))
    F, P = len(d["ext"]), len(d["pts0"])
    fi, pi, obs = d["fi"], d["pi"], d["obs"]
    cams = bo.frame_parameters(d["ext"]).reshape(F, 6)
    pb = ops.BADevice(d["K"], fi, pi, obs, F, P, DEV, default_context())
    cd, pd = dev(cams), dev(d["pts0"])
    Jc, Jp = (a.cpu().numpy() for a in pb.jacobian(cd, pd))
    res = pb.residual(cd, pd, True)[1].cpu().numpy()
    x0 = np.hstack([cams.ravel(), d["pts0"].ravel()])
    np.testing.assert_allclose(res.ravel(), bo.point_fun(x0, d["K"], F, P, fi, pi, obs), rtol=1e-9, atol=1e-7)
    rng = np.random.default_rng(0)
    for o in rng.choice(len(fi), 40, replace=False):      # Jacobian blocks of sampled observations, oracle central differences
        f, p_ = int(fi[o]), int(pi[o])
        one = lambda x: bo.point_fun(x, d["K"], F, P, fi[o:o + 1], pi[o:o + 1], obs[o:o + 1])
        for base, width, blk in ((6 * f, 6, Jc[o]), (6 * F + 3 * p_, 3, Jp[o])):
            for k in range(width):
                h = 1e-6 * max(1.0, abs(x0[base + k]))
                e = np.zeros_like(x0)
                e[base + k] = h
                fd = (one(x0 + e) - one(x0 - e)) / (2 * h)
                np.testing.assert_allclose(blk[:, k], fd, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(blk).max()))
    Bo, gco, Co, gpo = _dense_normal({"fi": fi, "pi": pi}, F, P, Jc, Jp, res)
    reg = 1e-3
    Bd_o = Bo + reg * np.einsum("fii,ij->fij", Bo, np.eye(6))
    Cd_o = Co + reg * np.einsum("pii,ij->pij", Co, np.eye(3))
    Cinv_o = np.linalg.inv(Cd_o)
    E = np.einsum("omi,omj->oij", Jc, Jp)                                  # [O, 6, 3]
    So = np.zeros((6 * F, 6 * F))
    vo = gco.reshape(-1).copy()
    for f in range(F):
        So[6 * f:6 * f + 6, 6 * f:6 * f + 6] = Bd_o[f]
    order = np.argsort(pi, kind="stable")
    starts = np.searchsorted(pi[order], np.arange(P + 1))
    for p_ in range(P):
        oo_ = order[starts[p_]:starts[p_ + 1]]
        if len(oo_) == 0:
            continue
        Ep = np.zeros((6 * F, 3))
        for o in oo_:
            Ep[6 * fi[o]:6 * fi[o] + 6] += E[o]
        rows = np.unique(fi[oo_])
        idx = (6 * rows[:, None] + np.arange(6)).ravel()
        So[np.ix_(idx, idx)] -= Ep[idx] @ Cinv_o[p_] @ Ep[idx].T
        vo[idx] -= Ep[idx] @ (Cinv_o[p_] @ gpo[p_])
    dc_o = np.linalg.solve(So, vo)
    dp_o = np.empty((P, 3))
    for p_ in range(P):
        oo_ = order[starts[p_]:starts[p_ + 1]]
        t = gpo[p_].copy()
        for o in oo_:
            t -= E[o].T @ dc_o[6 * fi[o]:6 * fi[o] + 6]
        dp_o[p_] = Cinv_o[p_] @ t
    B, gc, C, gp = pb.normal_eq(cd, pd)
    np.testing.assert_allclose(B.cpu().numpy(), Bo, rtol=1e-10, atol=1e-9 * np.abs(Bo).max())
    Bd = B.clone()
    Bd.diagonal(dim1=1, dim2=2).add_(reg * torch.diagonal(B, dim1=1, dim2=2))
    Cd = C.clone()
    Cd[:, [0, 3, 5]] += reg * C[:, [0, 3, 5]]
    S, v, Cinv = pb.schur(cd, pd, Bd, Cd, gc, gp)
    Sl = np.tril(S.cpu().numpy())
    np.testing.assert_allclose(Sl + np.tril(Sl, -1).T, So, rtol=1e-8, atol=1e-9 * np.abs(So).max())
    assert int(ops.chol_solve(S, v, half_bandwidth=6 * pb.cam_span + 5)) == 0
    dc = v.reshape(F, 6)
    dp = pb.backsub(cd, pd, Cinv, gp, dc).cpu().numpy()
    np.testing.assert_allclose(dc.cpu().numpy().ravel(), dc_o, rtol=1e-6, atol=1e-8 * np.abs(dc_o).max())
    np.testing.assert_allclose(dp, dp_o, rtol=1e-5, atol=1e-7 * np.abs(dp_o).max())


@pytest.mark.parametrize("case", ["a", "c", "d", "real", "tiny"])
def test_library_trf_driver_equals_python_driver_bitwise(golden_dir, case, monkeypatch):
    """mm_ba_trf (the loop inside the library, csrc/trf.hip) and the Python-sequenced loop issue the same kernels with
    the same scalars: identical nfev / status / cost / x to the last bit, and the same verbose=2 table."""
    if case == "real":
        d = np.load(os.path.join(golden_dir, "o1_real_match_ba.npz"))
        pr = {k: d[k] for k in ("ext", "K", "pts0", "obs", "fi", "pi")}
        kw = {}
    elif case == "tiny":
        pr = synth.make_ba_problem(3, 12, 3, seed=4)        # smaller than one Cholesky block
        kw = dict(ftol=1e-10, xtol=1e-10, gtol=1e-10, max_nfev=7)      # ... stopped by max_nfev
    else:
        d = np.load(os.path.join(golden_dir, f"g5_adjust_points_{case}.npz"))
        pr = synth.make_ba_problem(*(int(d[k]) for k in ("F", "P", "L")), seed=int(d["seed"]))
        kw = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=60) if case == "c" else {}
    out = {}
    for drv in ("python", "library"):
        monkeypatch.setenv("MM_TRF_DRIVER", drv)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], verbose=2, **kw)
        out[drv] = (res, buf.getvalue())
        assert ("library" in res.host_segments_ms) == (drv == "library")
    a, b = out["python"][0], out["library"][0]
    assert (a.nfev, a.njev, a.status, a.iterations) == (b.nfev, b.njev, b.status, b.iterations)
    assert a.cost == b.cost and a.optimality == b.optimality
    assert np.array_equal(a.x, b.x)
    assert out["python"][1] == out["library"][1]
    if case == "tiny":
        assert a.status == 0 and a.nfev == 7


def test_trf_folded_launches_equal_the_separate_ones(golden_dir):
    """Inside mm_ba_trf the damping, the band zero fill of S and the factorisation's fills ride in schur_prepare_kernel
    (mm_ba_schur_solve_damped); MM_SCHUR_FOLD=0 keeps mm_ba_damp + fill + mm_ba_schur_solve + chol_init_kernel as separate
    launches (what the Python-sequenced loop and the sharded loop issue).  Same arithmetic: identical iterates, bit for bit.
    (The switch is read once per process: the unfolded run is a fresh interpreter.)"""
    import subprocess
    import sys
    d = np.load(os.path.join(golden_dir, "g5_adjust_points_d.npz"))       # 120 cameras: pair list + single-launch factorisation
    F, P, L, seed = (int(d[k]) for k in ("F", "P", "L", "seed"))
    pr = synth.make_ba_problem(F, P, L, seed=seed)
    res = bundleAdjuster.solvePoints(pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"], verbose=0)
    assert "library" in res.host_segments_ms
    code = (
        "import numpy as np\n"
        "from meatmodeler_amd import synth, bundleAdjuster\n"
        f"pr = synth.make_ba_problem({F}, {P}, {L}, seed={seed})\n"
        "r = bundleAdjuster.solvePoints(pr['ext'], pr['K'], pr['pts0'], pr['obs'], pr['fi'], pr['pi'], verbose=0)\n"
        "assert 'library' in r.host_segments_ms\n"
        "print('RESULT', r.nfev, r.status, repr(float(r.cost)), repr(float(np.asarray(r.x, dtype=np.float64).sum())))\n")
    env = dict(os.environ, MM_SCHUR_FOLD="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1].split()
    assert (int(line[1]), int(line[2])) == (res.nfev, res.status)
    assert float(line[3]) == float(res.cost)
    assert float(line[4]) == float(np.asarray(res.x, dtype=np.float64).sum())


@pytest.mark.parametrize("driver", ["library", "python"])
def test_trf_survives_an_abandoned_factorisation(golden_dir, driver, monkeypatch):
    """info = -1 (the single-launch banded factorisation gave up waiting: its workgroups were not co-resident) is not an
    error any more: the attempt is repeated with the launch-per-column factorisation and the solve continues there.  The
    abandonment is provoked through the context's test hook; the result equals the undisturbed run to rounding (the two
    factorisations order their sums differently)."""
    _needs_single_launch_chol()
    from meatmodeler_amd._lib import default_context
    monkeypatch.setenv("MM_TRF_DRIVER", driver)
    ctx = default_context()
    d = np.load(os.path.join(golden_dir, "g5_adjust_points_d.npz"))       # 120 cameras: the band path
    pr = synth.make_ba_problem(*(int(d[k]) for k in ("F", "P", "L")), seed=int(d["seed"]))
    args = (pr["ext"], pr["K"], pr["pts0"], pr["obs"], pr["fi"], pr["pi"])
    ref = bundleAdjuster.solvePoints(*args, verbose=0)
    assert ctx.control(ctx.CTL_CHOL_LAST_PATH) == 1
    try:
        ctx.control(ctx.CTL_CHOL_FORCE_ABANDON, 1)
        res = bundleAdjuster.solvePoints(*args, verbose=0)
    finally:
        ctx.control(ctx.CTL_CHOL_FORCE_ABANDON, 0)
        ctx.control(ctx.CTL_CHOL_AVOID_FUSED, 0)
    if driver == "library":
        assert res.chol_fallbacks == 1
    if driver == "python":
        ctx.sync()                                          # (the Python-sequenced loop never calls mm_ctx_sync itself)
    assert ctx.control(ctx.CTL_CHOL_RESERVED) == 0          # the budget share went back with the end of the solve
    assert (res.nfev, res.status) == (ref.nfev, ref.status)
    assert abs(res.cost - ref.cost) <= 1e-9 * ref.cost
    np.testing.assert_allclose(res.x, ref.x, rtol=1e-6, atol=1e-8)
    # and the next solve is back on the single-launch path
    bundleAdjuster.solvePoints(*args, verbose=0)
    assert ctx.control(ctx.CTL_CHOL_LAST_PATH) == 1


def test_chol_single_launch_budget_over_contexts():
    """The workgroups of the single-launch factorisation must all be resident (one per compute unit).  Contexts that
    solve at the same time share a per-process budget of the device's compute units: a launch that does not fit takes
    the launch-per-column path (same solution).  A share returns at the owner's next synchronisation -- or, since round 4
    (ADVICE), without one: an event recorded behind the solve lets ANOTHER context that finds the budget spent collect the
    shares of solves that have finished (the Python-sequenced driver and direct callers never call mm_ctx_sync)."""
    _needs_single_launch_chol()
    from meatmodeler_amd._lib import Context
    n, hb = 3000, 528                      # two-ended grid: 2 * 46 + 45 = 137 workgroups
    rng = np.random.default_rng(8)
    M = np.tril(np.triu(rng.normal(size=(n, n)), -hb // 2))
    A = M @ M.T + n * np.eye(n)
    b = rng.normal(size=n)
    ref = np.linalg.solve(A, b)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    ctxs = [Context(DEV, s) for s in streams]
    cus = ctxs[0].control(ctxs[0].CTL_CU_COUNT)
    assert cus >= 64
    fit = cus // 137                       # launches that fit side by side (1 on a 256-CU MI355X)
    if fit != 1:
        pytest.skip("written for one 137-workgroup grid per device")
    ins = [(dev(A), dev(b)) for _ in range(4)]
    torch.cuda.synchronize()
    sols, paths = [], []
    for k, (c, st) in enumerate(zip(ctxs, streams)):
        with torch.cuda.stream(st):
            if k == 0:
                torch.cuda._sleep(400_000_000)      # ~0.2 s in front of the first solve: it is still in flight below
            Ad, bd = ins[k]
            info = ops.chol_solve_sym(Ad, bd, c, half_bandwidth=hb, both_triangles=True)
        sols.append((Ad, bd, info))
        paths.append(c.control(c.CTL_CHOL_LAST_PATH))
    assert paths == [1, 0, 0]
    assert [c.control(c.CTL_CHOL_RESERVED) for c in ctxs] == [137, 0, 0]
    torch.cuda.synchronize()               # (the device is idle; NO mm_ctx_sync: context 0 still holds its share)
    assert ctxs[0].control(ctxs[0].CTL_CHOL_RESERVED) == 137
    for Ad, bd, info in sols:
        assert int(info) == 0
        np.testing.assert_allclose(bd.cpu().numpy(), ref, rtol=1e-9, atol=1e-12)
    # the last context finds the budget spent, sees that context 0's solve has finished, collects its share and fits
    with torch.cuda.stream(streams[2]):
        Ad, bd = ins[3]
        ops.chol_solve_sym(Ad, bd, ctxs[2], half_bandwidth=hb, both_triangles=True)
    assert ctxs[2].control(ctxs[2].CTL_CHOL_LAST_PATH) == 1
    assert [c.control(c.CTL_CHOL_RESERVED) for c in ctxs] == [0, 0, 137]
    ctxs[2].sync()
    assert ctxs[2].control(ctxs[2].CTL_CHOL_RESERVED) == 0      # ... and a synchronisation still returns it
    np.testing.assert_allclose(bd.cpu().numpy(), ref, rtol=1e-9, atol=1e-12)


def test_adjust_points_raises_on_non_finite():
    pr = synth.make_ba_problem(4, 10, 3, seed=1)
    pts = pr["pts0"].copy()
    pts[0, 0] = np.nan
    with pytest.raises(ValueError):
        bundleAdjuster.solvePoints(pr["ext"], pr["K"], pts, pr["obs"], pr["fi"], pr["pi"], verbose=0)


def test_adjust_pose_vs_reference_golden(golden_dir):
    """adjustPose (pose-only, dense 'exact' trust region in the reference) against golden G6."""
    d = np.load(os.path.join(golden_dir, "g6_adjust_pose.npz"))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bundleAdjuster.adjustPose(d["ext0"], d["K"], d["obs"])
    table = buf.getvalue().splitlines()
    ref_table = open(os.path.join(golden_dir, "g6_adjust_pose_table.txt")).read().splitlines()
    assert len(out) == len(d["ext0"]) and out[0].shape == (3, 4)
    # well-conditioned problem (12 fixed points per camera): the iterates follow SciPy's closely
    assert len(table) == len(ref_table)
    assert [l.split()[:3] for l in table[1:-2]] == [l.split()[:3] for l in ref_table[1:-2]]   # iteration, nfev, cost
    np.testing.assert_allclose(np.array(out), d["result"], rtol=1e-6, atol=1e-7)


def test_point_fun_and_project_surface(golden_dir):
    d = np.load(os.path.join(golden_dir, "g1_rotate_project.npz"))
    np.testing.assert_allclose(bundleAdjuster.project(d["pts"], d["params"], d["K"]), d["projected"], rtol=1e-10,
                               atol=1e-8)
    np.testing.assert_allclose(bundleAdjuster.rotate(d["pts"], d["params"][:, :3]), d["rotated"], rtol=1e-12, atol=1e-12)
    g3 = np.load(os.path.join(golden_dir, "g3_point_pose_fun.npz"))
    F = 4
    fi = np.repeat(np.arange(F), 12)
    pi = np.tile(np.arange(12), F)
    r = bundleAdjuster.poseFun(g3["pose_cams"], g3["pose_K"], F, fi, pi, g3["pose_pts3"], g3["pose_obs"])
    np.testing.assert_allclose(r, g3["pose_res"], rtol=1e-10, atol=1e-9)


# ============================================================================================== drop-in flow

@pytest.mark.parametrize("n_frames,arc", [(4, 4.5), (20, 30.0)])
def test_processor_drop_in_flow_matches_oracle_flow(n_frames, arc):
    """featureTracking -> pointTracking -> triangulatePoints -> managePoints (-> adjustPoints) against the same flow
    built from the oracle's functions: on a 4-frame clip, and on BASELINE config 1 at its stated size (20 frames
    640 x 480 on a 30 degree orbit arc) through the per-keyframe drop-in surface."""
    frames, ext, K = synth.render_orbit_frames(n_frames, 640, 480, arc_deg=arc)
    orb = processor.ORB_create(nfeatures=800)
    pts_prev, desc_prev = orb.detectAndCompute(frames[0], None)
    o_prev = oo.detect_compute(frames[0], 800, brief_pattern())
    np.testing.assert_array_equal(np.asarray(desc_prev), o_prev["desc"])
    assert pts_prev[3].pt == (float(o_prev["xy"][3, 0]), float(o_prev["xy"][3, 1]))
    tracks, otracks, popped, opopped = [], [], [], []
    for k in range(1, n_frames):
        pm, cm, pts_new, desc_new = processor.featureTracking(frames[k], pts_prev, desc_prev, orb, dict(algorithm=6))
        o_new = oo.detect_compute(frames[k], 800, brief_pattern())
        io_, do_ = oo.bf_knn2(o_prev["desc"], o_new["desc"])
        good = oo.ratio_filter(io_, do_, 0.75)
        opm = o_prev["xy"][good[:, 0]].astype(np.float64)
        ocm = o_new["xy"][good[:, 1]].astype(np.float64)
        np.testing.assert_array_equal(pm, opm)
        np.testing.assert_array_equal(cm, ocm)
        assert pm.dtype == np.float64 and len(pm) > 100
        p, tracks = processor.pointTracking(tracks, k - 1, pm, k, cm)
        op, otracks = bo.point_tracking(otracks, k - 1, opm, k, ocm)
        popped += p
        opopped += op
        pts_prev, desc_prev, o_prev = pts_new, desc_new, o_new
    final, ofinal = popped + tracks, opopped + otracks
    assert len(final) == len(ofinal) and len(final) > 100
    proj = [K @ e for e in ext]
    processor.triangulatePoints(final, proj)
    for t, ot in zip(final, ofinal):
        assert list(t.getCoordinates().items()) == list(ot.getCoordinates().items())
    f0 = np.array([t.getTriangulationData()[0] for t in ofinal])
    f1 = np.array([t.getTriangulationData()[1] for t in ofinal])
    x0 = np.array([t.getTriangulationData()[2] for t in ofinal])
    x1 = np.array([t.getTriangulationData()[3] for t in ofinal])
    Xo = bo.triangulate_dlt(np.array(proj)[f0], np.array(proj)[f1], x0, x1)
    X = np.concatenate([t.getPoint() for t in final])
    assert final[0].getPoint().shape == (1, 3)
    ok = np.isfinite(Xo).all(1) & (np.abs(Xo).max(1) < 1e3)
    np.testing.assert_allclose(X[ok], Xo[ok], rtol=1e-6, atol=1e-7)
    points, coords, fidx, pidx = processor.managePoints(final)
    for i, t in enumerate(ofinal):
        t.setPoint(final[i].getPoint())
    op_, oc_, of_, opi_ = bo.manage_points(ofinal)
    assert coords == oc_ and fidx == of_ and pidx == opi_ and np.array(points).shape == (len(final), 1, 3)
    if n_frames < 20:
        return
    # config 1 end to end: the bundle adjustment of the reference's tail (processor.py:463-470) on these tracks.  The
    # matches carry outliers (no RANSAC, as in the reference), so the trust-region path is chaotic: what is asserted is
    # that the solver's bookkeeping agrees with the ORACLE's cost function at its result and that it ends well below
    # the start (the SciPy recipe itself needs minutes on such data; tests/golden/o1_real_match_ba.npz pins one such run).
    ext34 = np.asarray(ext)[:, :3, :]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pts_ba, ext_ba = bundleAdjuster.adjustPoints(ext34, K, np.array(points), np.array(coords), np.array(fidx), np.array(pidx))
    F, P = n_frames, len(final)
    x0 = np.hstack([bo.frame_parameters(ext34), np.array(points).reshape(-1)])
    cams = np.array([np.concatenate([bo.frame_parameters(e[None, :3])[:3], e[:3, 3]]) for e in ext_ba])
    x1 = np.hstack([cams.ravel(), pts_ba.ravel()])
    c0 = 0.5 * np.sum(bo.point_fun(x0, K, F, P, np.array(fidx), np.array(pidx), np.array(coords)) ** 2)
    c1 = 0.5 * np.sum(bo.point_fun(x1, K, F, P, np.array(fidx), np.array(pidx), np.array(coords)) ** 2)
    last = [l for l in buf.getvalue().splitlines() if l.startswith("Function evaluations")][-1]
    assert abs(float(last.split("final cost ")[1].split(",")[0]) - c1) <= 2e-4 * c1
    assert np.isfinite(c1) and c1 < 0.5 * c0 and pts_ba.shape == (P, 3) and len(ext_ba) == F


def _needs_single_launch_chol():
    if os.environ.get("MM_CHOL_FUSED") == "0":
        pytest.skip("MM_CHOL_FUSED=0 forces the launch-per-column factorisation: this test is about the single launch")


def _needs_library_driver():
    if os.environ.get("MM_TRF_DRIVER", "library") != "library":
        pytest.skip("MM_TRF_DRIVER selects the Python-sequenced loop: this test is about the loop inside the library")


def _link_both(kp_count, kp_xy, mc, mm):
    """(host linker result, device linker result) on the same tables; matches padded to the key point capacity."""
    F, cap = kp_xy.shape[0], kp_xy.shape[1]
    pad = np.zeros((max(F - 1, 0), cap, 2), np.int32)
    pad[:, :mm.shape[1]] = mm[:, :cap]
    host = ClipPipeline.link(None, kp_count, kp_xy, mc, pad)
    tp, of, ok, bad = ops.link_tracks_device(dev(kp_count.astype(np.int32)), dev(kp_xy.astype(np.float32)),
                                             dev(mc.astype(np.int32)), dev(pad))
    assert not bad
    return host, (tp.cpu().numpy(), of.cpu().numpy(), ok.cpu().numpy())


def test_link_tracks_device_golden(golden_dir):
    """mm_link_tracks_device == the reference's per-call pointTracking + managePoints (G7 scripts)."""
    for sc in json.load(open(os.path.join(golden_dir, "g7_point_tracking.json"))):
        kp = {int(k): v for k, v in sc["kp"].items()}
        matches = {int(k): v for k, v in sc["matches"].items()}
        F = len(kp)
        cap = max(max(len(v) for v in kp.values()), max(len(v) for v in matches.values()))
        kp_xy = np.zeros((F, cap, 2), np.float32)
        kp_count = np.zeros(F, np.int32)
        for f, v in kp.items():
            kp_xy[f, :len(v)] = v
            kp_count[f] = len(v)
        mm = np.zeros((F - 1, cap, 2), np.int32)
        mc = np.zeros(F - 1, np.int32)
        for f, v in matches.items():
            mm[f, :len(v)] = v
            mc[f] = len(v)
        _, (tp, of, ok) = _link_both(kp_count, kp_xy, mc, mm)
        coords, fi, pi = ClipPipeline.flatten(tp, of, ok, kp_xy)
        mg = sc["manage"]
        assert coords.tolist() == mg["coordinates"]
        assert fi.tolist() == mg["frame_indices"] and pi.tolist() == mg["point_indices"]
        assert len(tp) - 1 == mg["points_shape"][0]


@pytest.mark.parametrize("F,nk,frac,seed", [(9, 120, 0.6, 1), (40, 1500, 0.8, 2), (12, 4000, 0.95, 3), (6, 8192, 0.9, 4)])
def test_link_tracks_device_equals_host_linker(F, nk, frac, seed):
    """Random clips with duplicate coordinates (several key points at one pixel), several matches landing on one
    track, ragged key point counts and -0.0 coordinates: device CSR == host CSR, entry by entry."""
    rng = np.random.default_rng(seed)
    kp_xy = (np.round(rng.uniform(0, 60 if nk > 1000 else 300, (F, nk, 2)) * 2) / 2).astype(np.float32)
    kp_xy[:, 7] = kp_xy[:, 3]
    kp_xy[:, 11, 0] = -0.0
    kp_xy[:, 12, 0] = 0.0
    kp_xy[:, 12, 1] = kp_xy[:, 11, 1]
    kp_count = np.full(F, nk, np.int32)
    kp_count[F // 2] = nk - 20
    mc = np.zeros(F - 1, np.int32)
    mm = np.zeros((F - 1, nk, 2), np.int32)
    for f in range(F - 1):
        q = np.sort(rng.choice(kp_count[f], size=int(kp_count[f] * frac), replace=False))
        t = rng.integers(0, kp_count[f + 1], size=q.size)
        mc[f] = q.size
        mm[f, :q.size, 0] = q
        mm[f, :q.size, 1] = t
    if F > 10:
        mc[5] = 0           # a pair without matches pops every live track
    (tp0, of0, ok0), (tp1, of1, ok1) = _link_both(kp_count, kp_xy, mc, mm)
    assert np.array_equal(tp0, tp1.astype(np.int64))
    assert np.array_equal(of0, of1) and np.array_equal(ok0, ok1)
    assert len(tp0) - 1 > 0


@pytest.mark.parametrize("variant", ["parallel", "serial"])
def test_link_tracks_device_formulations_agree_on_degenerate_clips(variant, monkeypatch):
    """Both formulations of the device linker (pointer doubling over the whole clip / one workgroup walking the pairs)
    against the host linker where the has-a-track question chains through nodes with several matches: every key point of the
    clip at ONE coordinate; two coordinates alternating; queries repeated inside a pair."""
    monkeypatch.setenv("MM_LINK_VARIANT", variant)
    rng = np.random.default_rng(5)
    F, nk = 24, 96
    for mode in range(3):
        kp_xy = np.zeros((F, nk, 2), np.float32)
        if mode == 1:
            kp_xy[:, :, 0] = (np.arange(nk) % 2)[None, :]
        if mode == 2:
            kp_xy[:, :, 0] = (np.arange(nk) // 7)[None, :]
        kp_count = np.full(F, nk, np.int32)
        mc = np.zeros(F - 1, np.int32)
        mm = np.zeros((F - 1, nk, 2), np.int32)
        for f in range(F - 1):
            m = int(rng.integers(0, nk))
            mc[f] = m
            mm[f, :m, 0] = rng.integers(0, nk, size=m)          # repeated queries allowed
            mm[f, :m, 1] = rng.integers(0, nk, size=m)
        (tp0, of0, ok0), (tp1, of1, ok1) = _link_both(kp_count, kp_xy, mc, mm)
        assert np.array_equal(tp0, tp1.astype(np.int64)), mode
        assert np.array_equal(of0, of1) and np.array_equal(ok0, ok1), mode


def test_link_tracks_device_safety_valve_takes_the_serial_formulation(monkeypatch):
    """ADVICE / VERDICT round 3: a clip whose key points all share one coordinate makes every match 'uncertain' and the
    parallel formulation's one-workgroup fixed-point pass run as many rounds as the clip is long.  Past a work budget
    (rounds x uncertain matches) the call now switches to the serial formulation by itself: same CSR as the host linker, and
    the context says which formulation ran.  An ordinary clip stays on the parallel formulation."""
    from meatmodeler_amd._lib import default_context
    monkeypatch.delenv("MM_LINK_VARIANT", raising=False)
    ctx = default_context()
    rng = np.random.default_rng(6)
    F, nk = 40, 256
    kp_xy = np.zeros((F, nk, 2), np.float32)                # every key point of the clip at (0, 0)
    kp_count = np.full(F, nk, np.int32)
    mc = np.zeros(F - 1, np.int32)
    mm = np.zeros((F - 1, nk, 2), np.int32)
    for f in range(F - 1):
        m = int(rng.integers(nk // 2, nk))
        mc[f] = m
        mm[f, :m, 0] = rng.integers(0, nk, size=m)
        mm[f, :m, 1] = rng.integers(0, nk, size=m)
    monkeypatch.setenv("MM_LINK_UNCERTAIN_BUDGET", "2000")     # ~170 uncertain matches per pair x 39 pairs: over at once
    (tp0, of0, ok0), (tp1, of1, ok1) = _link_both(kp_count, kp_xy, mc, mm)
    assert ctx.control(ctx.CTL_LINK_LAST_VARIANT) == 0
    assert np.array_equal(tp0, tp1.astype(np.int64)) and np.array_equal(of0, of1) and np.array_equal(ok0, ok1)
    monkeypatch.delenv("MM_LINK_UNCERTAIN_BUDGET")              # default budget: this small clip fits, parallel again
    (tp0, of0, ok0), (tp2, of2, ok2) = _link_both(kp_count, kp_xy, mc, mm)
    assert ctx.control(ctx.CTL_LINK_LAST_VARIANT) == 1
    assert np.array_equal(tp1, tp2) and np.array_equal(of1, of2) and np.array_equal(ok1, ok2)
    monkeypatch.setenv("MM_LINK_VARIANT", "serial")
    _link_both(kp_count, kp_xy, mc, mm)
    assert ctx.control(ctx.CTL_LINK_LAST_VARIANT) == 0


def test_link_tracks_device_serial_equals_parallel(monkeypatch):
    """The two device formulations on a clip-sized case with duplicate coordinates: identical CSR."""
    rng = np.random.default_rng(8)
    F, nk = 120, 3000
    kp_xy = (np.round(rng.uniform(0, 90, (F, nk, 2)) * 2) / 2).astype(np.float32)
    kp_count = rng.integers(nk - 500, nk + 1, size=F).astype(np.int32)
    mc = np.zeros(F - 1, np.int32)
    mm = np.zeros((F - 1, nk, 2), np.int32)
    for f in range(F - 1):
        m = int(min(kp_count[f], kp_count[f + 1]) * 0.8)
        mc[f] = m
        mm[f, :m, 0] = np.sort(rng.choice(kp_count[f], size=m, replace=False))
        mm[f, :m, 1] = rng.integers(0, kp_count[f + 1], size=m)
    res = {}
    for variant in ("parallel", "serial"):
        monkeypatch.setenv("MM_LINK_VARIANT", variant)
        tp, of, ok, bad = ops.link_tracks_device(dev(kp_count), dev(kp_xy), dev(mc), dev(mm))
        assert not bad
        res[variant] = (tp.cpu().numpy(), of.cpu().numpy(), ok.cpu().numpy())
    for a, b in zip(res["parallel"], res["serial"]):
        assert np.array_equal(a, b)
    assert len(res["serial"][0]) > 1000


def test_link_tracks_device_full_size_properties():
    """BASELINE size (500 frames x 4000 key points, ~3000 matches per pair, distinct coordinates): properties that hold
    for any correct linking -- every track has >= 2 observations on consecutive ascending frames, every step of a track
    is one of the pair's matches, every match is used exactly once (as a continuation or as the start of a track), no
    (frame, key point) starts or continues two tracks from the same match -- and the device result equals the host
    linker's."""
    rng = np.random.default_rng(11)
    F, nk = 500, 4000
    # distinct coordinates per frame: canonical key point == key point
    kp_xy = np.stack([np.tile(np.arange(nk, dtype=np.float32), (F, 1)), np.tile(np.arange(F, dtype=np.float32)[:, None], (1, nk))], -1)
    kp_count = np.full(F, nk, np.int32)
    mc = np.zeros(F - 1, np.int32)
    mm = np.zeros((F - 1, nk, 2), np.int32)
    for f in range(F - 1):
        m = int(rng.integers(2500, 3500))
        q = np.sort(rng.choice(nk, size=m, replace=False))       # a query key point matches at most once (kNN + ratio)
        t = rng.integers(0, nk, size=m)                          # several queries may hit the same train key point
        mc[f], mm[f, :m, 0], mm[f, :m, 1] = m, q, t
    (tp0, of0, ok0), (tp, of, ok) = _link_both(kp_count, kp_xy, mc, mm)
    assert np.array_equal(tp0, tp.astype(np.int64)) and np.array_equal(of0, of) and np.array_equal(ok0, ok)
    lens = np.diff(tp)
    assert lens.min() >= 2 and tp[-1] == len(of)
    inner = np.ones(len(of), bool)
    inner[tp[1:-1]] = False                                      # positions that continue the previous observation
    inner[0] = False
    step_f, step_a, step_b = of[:-1][inner[1:]], ok[:-1][inner[1:]], ok[1:][inner[1:]]
    assert np.array_equal(of[1:][inner[1:]], step_f + 1)         # consecutive frames inside a track
    key = lambda f, a, b: (f.astype(np.int64) * nk + a) * nk + b
    all_matches = np.concatenate([key(np.full(mc[f], f), mm[f, :mc[f], 0], mm[f, :mc[f], 1]) for f in range(F - 1)])
    steps = key(step_f, step_a, step_b)
    assert len(np.unique(steps)) == len(steps)
    assert np.isin(steps, all_matches).all() and len(steps) == len(all_matches)   # every match is exactly one step


def test_link_tracks_device_empty_and_malformed():
    z = np.zeros
    tp, of, ok, bad = ops.link_tracks_device(dev(z(1, np.int32)), dev(z((1, 4, 2), np.float32)), dev(z(0, np.int32)),
                                             dev(z((0, 4, 2), np.int32)))
    assert tp.cpu().tolist() == [0] and of.numel() == 0 and not bad
    tp, of, ok, bad = ops.link_tracks_device(dev(np.array([3, 3], np.int32)), dev(z((2, 4, 2), np.float32)),
                                             dev(z(1, np.int32)), dev(z((1, 4, 2), np.int32)))
    assert tp.cpu().tolist() == [0] and of.numel() == 0 and not bad
    mm = z((1, 4, 2), np.int32)
    mm[0, 1] = (3, 0)       # query index 3 >= kp_count 3
    tp, of, ok, bad = ops.link_tracks_device(dev(np.array([3, 3], np.int32)), dev(z((2, 4, 2), np.float32)),
                                             dev(np.array([2], np.int32)), dev(mm))
    assert bad and tp.cpu().tolist() == [0, 2]


def test_clip_pipeline_equals_per_keyframe_drop_in():
    """The batched pipeline (detect all / match all / mm_link_tracks_clip) produces the same tracks as the per-keyframe
    drop-in functions."""
    frames, ext, K = synth.render_orbit_frames(5, 640, 480, arc_deg=6.0)
    pipe = ClipPipeline(480, 640, 800, batch=3)
    out = pipe.run(dev(frames), K, ext, ba=False)
    orb = processor.ORB_create(nfeatures=800)
    pts_prev, desc_prev = orb.detectAndCompute(frames[0], None)
    tracks, popped = [], []
    for k in range(1, 5):
        pm, cm, pts_prev, desc_prev = processor.featureTracking(frames[k], pts_prev, desc_prev, orb, None)
        p, tracks = processor.pointTracking(tracks, k - 1, pm, k, cm)
        popped += p
    final = popped + tracks
    assert out["n_tracks"] == len(final)
    xy = out["det"]["xy"].cpu().numpy()
    ClipPipeline.tracks_to_host(out)
    tp, of, ok = out["track_ptr"], out["obs_frame"], out["obs_kp"]
    for i, t in enumerate(final):
        got = [(int(of[j]), (float(xy[of[j], ok[j], 0]), float(xy[of[j], ok[j], 1]))) for j in range(tp[i], tp[i + 1])]
        want = [(int(f), (float(c[0]), float(c[1]))) for f, c in t.getCoordinates().items()]
        assert got == want, i


def test_sliding_window_ba_equals_window_by_window_adjustment():
    """ClipPipeline.adjust_windows (the reference's commented incremental hook, processor.py:395-408, bounded to a
    window): the device-side selection / flattening of every window equals a plain NumPy selection in managePoints
    order, every window is adjusted by exactly the adjustPoints solver (same nfev and cost, bit for bit), later windows
    start from the written-back cameras and points, and the cost the oracle computes at a window's result is the
    reported one."""
    from meatmodeler_amd.bundleAdjuster import SchurTRF, frameParameters
    F, W, S = 9, 5, 2
    frames, ext, K = synth.render_orbit_frames(F, 640, 480, arc_deg=10.0)
    pipe = ClipPipeline(480, 640, 600, batch=F)
    out = pipe.run(dev(frames), K, ext, ba=False)
    res = pipe.adjust_windows(out, K, ext, window=W, stride=S)
    ClipPipeline.tracks_to_host(out)
    tp, of, ok = out["track_ptr"], out["obs_frame"], out["obs_kp"]
    xy = out["xy_dev"].cpu().numpy()
    cams = frameParameters(np.asarray(ext)[:, :3, :]).reshape(F, 6)
    pts = out["points0"].cpu().numpy().copy()
    first, last = of[tp[:-1]], of[tp[1:] - 1]
    wins = iter(res["windows"])
    n_checked = 0
    for hi in list(range(W, F, S)) + [F]:
        lo = max(0, hi - W)
        sel = np.nonzero(ClipPipeline.window_selection(first, last, lo, hi, F))[0]
        if len(sel) == 0:
            continue
        fi, pi, coords = [], [], []
        for j, t in enumerate(sel):
            for o in range(tp[t], tp[t + 1]):
                fi.append(of[o] - lo)
                pi.append(j)
                coords.append(xy[of[o], ok[o]])
        pb = ops.BADevice(K, np.array(fi), np.array(pi), np.array(coords, np.float64), hi - lo, len(sel), DEV)
        r = SchurTRF(pb).solve(dev(cams[lo:hi]), dev(pts[sel]))
        st = next(wins)
        assert (st["lo"], st["hi"], st["points"], st["observations"]) == (lo, hi, len(sel), len(fi))
        assert (st["nfev"], st["cost"]) == (r.nfev, r.cost)
        x1 = np.hstack([r.cams.cpu().numpy().ravel(), r.pts.cpu().numpy().ravel()])
        c_oracle = 0.5 * np.sum(bo.point_fun(x1, K, hi - lo, len(sel), np.array(fi), np.array(pi), np.array(coords)) ** 2)
        assert abs(c_oracle - st["cost"]) <= 1e-7 * max(c_oracle, 1.0)
        cams[lo:hi] = r.cams.cpu().numpy()
        pts[sel] = r.pts.cpu().numpy()
        n_checked += 1
    assert n_checked >= 2 and next(wins, None) is None
    assert np.array_equal(res["points"].cpu().numpy(), pts) and np.array_equal(res["cams"].cpu().numpy(), cams)


def test_wavefront_windows_in_flight_on_several_streams_equal_one_at_a_time():
    """order="wavefront": the windows of a pass share no camera and no point; solving four of them at once (one HIP
    stream, one library context and one host thread each) gives the result of solving them one after the other, bit for
    bit, and the same per-window nfev / cost."""
    F, W, S = 16, 4, 2
    frames, ext, K = synth.render_orbit_frames(F, 640, 480, arc_deg=16.0)
    pipe = ClipPipeline(480, 640, 600, batch=F)
    out = pipe.run(dev(frames), K, ext, ba=False)
    one = pipe.adjust_windows(out, K, ext, window=W, stride=S, order="wavefront")
    four = pipe.adjust_windows(out, K, ext, window=W, stride=S, order="wavefront", streams=4)
    assert len(one["windows"]) >= 6 and max(w["colour"] for w in one["windows"]) == 1
    assert one["windows"] == four["windows"]
    assert torch.equal(one["cams"], four["cams"]) and torch.equal(one["points"], four["points"])
    # (batched = True on 4-camera windows: reduced systems smaller than two Cholesky blocks are solved one by one inside
    # mm_ba_trf_batched -- same result; the lock-step path proper is tested on larger windows below)
    bat = pipe.adjust_windows(out, K, ext, window=W, stride=S, order="wavefront", batched=True)
    assert one["windows"] == bat["windows"]
    assert torch.equal(one["cams"], bat["cams"]) and torch.equal(one["points"], bat["points"])


def test_flatten_tracks_device_equals_manage_points(golden_dir):
    """mm_flatten_offsets / mm_flatten_tracks (managePoints behind the C ABI, reference processor.py:264-291): the flat
    observation arrays of all tracks equal the reference's golden G7 output and the NumPy flattening; a contiguous range
    (one rank's shard) and an index list with a frame offset (a sliding window) equal plain NumPy selections."""
    sc = json.load(open(os.path.join(golden_dir, "g7_point_tracking.json")))[0]
    kp = {int(k): v for k, v in sc["kp"].items()}
    matches = {int(k): v for k, v in sc["matches"].items()}
    F = len(kp)
    cap = max(max(len(v) for v in kp.values()), max(len(v) for v in matches.values()))
    kp_xy = np.zeros((F, cap, 2), np.float32)
    kp_count = np.zeros(F, np.int32)
    mm = np.zeros((F - 1, cap, 2), np.int32)
    mc = np.zeros(F - 1, np.int32)
    for f, v in kp.items():
        kp_xy[f, :len(v)] = v
        kp_count[f] = len(v)
    for f, v in matches.items():
        mm[f, :len(v)] = v
        mc[f] = len(v)
    tp, of, ok, bad = ops.link_tracks_device(dev(kp_count), dev(kp_xy), dev(mc), dev(mm))
    coords, fi, pi = ops.flatten_tracks(tp, of, ok, dev(kp_xy))
    mg = sc["manage"]
    assert coords.cpu().numpy().tolist() == mg["coordinates"]
    assert fi.cpu().numpy().tolist() == mg["frame_indices"] and pi.cpu().numpy().tolist() == mg["point_indices"]
    # a larger random problem: ranges and index lists against NumPy
    rng = np.random.default_rng(12)
    T, Fr, capr = 5000, 40, 300
    lens = rng.integers(2, 9, T)
    tpn = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ofn = np.concatenate([np.sort(rng.choice(Fr, l, replace=False)) for l in lens]).astype(np.int32)
    okn = rng.integers(0, capr, tpn[-1]).astype(np.int32)
    xyn = rng.normal(size=(Fr, capr, 2)).astype(np.float32)

    def ref(sel, off):
        c, f_, p_ = [], [], []
        for j, t in enumerate(sel):
            for o in range(tpn[t], tpn[t + 1]):
                c.append(xyn[ofn[o], okn[o]].astype(np.float64))
                f_.append(ofn[o] - off)
                p_.append(j)
        return np.array(c).reshape(-1, 2), np.array(f_, np.int32), np.array(p_, np.int32)
    d_tp, d_of, d_ok, d_xy = dev(tpn), dev(ofn), dev(okn), dev(xyn)
    cases = [dict(), dict(t_lo=1234, n_sel=2100), dict(sel=dev(np.sort(rng.choice(T, 1500, replace=False)).astype(np.int64)), frame_offset=7),
             dict(sel=dev(np.array([4999, 0, 17], np.int64))), dict(t_lo=10, n_sel=0)]
    for kw in cases:
        c, f_, p_ = ops.flatten_tracks(d_tp, d_of, d_ok, d_xy, **kw)
        if "sel" in kw:
            sel = kw["sel"].cpu().numpy()
        else:
            lo = kw.get("t_lo", 0)
            sel = np.arange(lo, lo + kw.get("n_sel", T - lo))
        rc, rf, rp = ref(sel, kw.get("frame_offset", 0))
        assert np.array_equal(c.cpu().numpy().reshape(-1, 2), rc) and np.array_equal(f_.cpu().numpy(), rf)
        assert np.array_equal(p_.cpu().numpy(), rp)


def test_c5_shape_4k_windows_on_one_gpu():
    """BASELINE config 5's shape on one GPU: 3840 x 2160 frames, 8000 key points per frame, sliding-window bundle
    adjustment with the wavefront schedule on 8 HIP streams (60 frames instead of 2000: the per-frame and per-window work
    is the full-size one).  Every window terminates on a tolerance (status > 0); with disjoint windows (stride = window)
    every window's reported cost is the oracle's cost function at the cameras / points it left behind (and no higher than
    where it started); with the usual half-overlapping windows eight windows in flight equal one at a time."""
    from meatmodeler_amd.bundleAdjuster import frameParameters
    F, W = 60, 20
    frames, ext, K = synth.render_orbit_frames_torch(F, 3840, 2160, DEV, arc_deg=0.36 * F, tex_size=4096)
    pipe = ClipPipeline(2160, 3840, 8000, batch=12)
    out = pipe.run(frames, K, ext, ba=False)
    del frames
    n_kp = out["det"]["n"].cpu().numpy()
    assert n_kp.min() > 6000 and out["n_tracks"] > 50000
    ClipPipeline.tracks_to_host(out)
    tp, of, ok = out["track_ptr"], out["obs_frame"], out["obs_kp"]
    xy = out["xy_dev"].cpu().numpy()
    first, last = of[tp[:-1]], of[tp[1:] - 1]
    cams0 = frameParameters(np.asarray(ext)[:, :3, :]).reshape(F, 6)
    pts0 = out["points0"].cpu().numpy()

    def window_problem(lo, hi):
        sel = np.nonzero(ClipPipeline.window_selection(first, last, lo, hi, F))[0]
        lens = tp[sel + 1] - tp[sel]
        oi = np.concatenate([np.arange(tp[t], tp[t + 1]) for t in sel])
        return sel, of[oi] - lo, np.repeat(np.arange(len(sel)), lens), xy[of[oi], ok[oi]].astype(np.float64)

    def cost(cams, pts, lo, hi, sel, fi, pi, coords):
        x = np.hstack([cams[lo:hi].ravel(), pts[sel].ravel()])
        return 0.5 * np.sum(bo.point_fun(x, K, hi - lo, len(sel), fi, pi, coords) ** 2)
    res = pipe.adjust_windows(out, K, ext, window=W, stride=W, order="wavefront", streams=8)
    cams, pts = res["cams"].cpu().numpy(), res["points"].cpu().numpy()
    assert len(res["windows"]) == 3
    for st in res["windows"]:
        assert st["status"] > 0 and st["points"] > 5000
        sel, fi, pi, coords = window_problem(st["lo"], st["hi"])
        assert (len(sel), len(fi)) == (st["points"], st["observations"])
        c = cost(cams, pts, st["lo"], st["hi"], sel, fi, pi, coords)
        assert abs(c - st["cost"]) <= 1e-7 * max(c, 1.0)
        assert c <= cost(cams0, pts0, st["lo"], st["hi"], sel, fi, pi, coords)
    # the usual half-overlapping windows: two colours; eight windows in flight give the result of one at a time, bit for bit
    res2 = pipe.adjust_windows(out, K, ext, window=W, stride=W // 2, order="wavefront", streams=8)
    assert len(res2["windows"]) == 5 and max(w["colour"] for w in res2["windows"]) == 1
    assert all(st["status"] > 0 and np.isfinite(st["cost"]) for st in res2["windows"])
    res1 = pipe.adjust_windows(out, K, ext, window=W, stride=W // 2, order="wavefront", streams=1)
    assert res1["windows"] == res2["windows"]
    assert torch.equal(res1["cams"], res2["cams"]) and torch.equal(res1["points"], res2["points"])
    # ... and so does advancing all windows of a pass in lock-step (mm_ba_trf_batched)
    res3 = pipe.adjust_windows(out, K, ext, window=W, stride=W // 2, order="wavefront", batched=True)
    assert res1["windows"] == res3["windows"]
    assert torch.equal(res1["cams"], res3["cams"]) and torch.equal(res1["points"], res3["points"])


def test_trf_batched_equals_one_problem_at_a_time():
    """mm_ba_trf_batched: independent problems of different sizes advanced in lock-step -- one launch of
    every kernel per round with blockIdx.y = problem -- against mm_ba_trf on each of them alone: identical cameras and
    points bit for bit, identical nfev / njev / status / iterations / cost / optimality.  The problems differ in size, in
    how many evaluations they need (a tight and a loose tolerance would not fit one call, so the spread comes from noise
    and outliers), in whether they stop on max_nfev, and one of them has observations no trial step improves at first."""
    _needs_single_launch_chol()
    _needs_library_driver()
    ctx = default_context()
    specs = [(24, 900, 6, 1, 0.5), (40, 2000, 6, 2, 1.0), (30, 1200, 5, 3, 3.0), (50, 4000, 8, 4, 0.3), (26, 700, 4, 5, 8.0),
             (120, 3000, 6, 6, 0.7)]
    probs, x0 = [], []
    for F, P, L, seed, noise in specs:
        pr = synth.make_ba_problem(F, P, L, seed=seed)
        rng = np.random.default_rng(seed)
        obs = pr["obs"] + rng.normal(0, noise, pr["obs"].shape)
        if seed == 5:      # gross outliers: rejected trial steps, many evaluations
            bad = rng.choice(len(obs), size=len(obs) // 50, replace=False)
            obs[bad] += rng.normal(0, 200.0, (len(bad), 2))
        with np.errstate(all="ignore"):
            cams0 = bundleAdjuster.frameParameters(pr["ext"]).reshape(F, 6)
        probs.append(ops.BADevice(pr["K"], pr["fi"], pr["pi"], obs, F, P, DEV, ctx))
        x0.append((dev(cams0), dev(pr["pts0"].copy())))
    for max_nfev in (None, 7):
        alone = []
        for pb, (c0, p0) in zip(probs, x0):
            c, p_ = c0.clone(), p0.clone()
            rep, _ = pb.trf_solve(c, p_, 1e-6, 1e-8, 1e-8, max_nfev=max_nfev)
            alone.append((c, p_, rep))
        cb, pbs_ = [c.clone() for c, _ in x0], [p_.clone() for _, p_ in x0]
        reps, solved_alone = ops.trf_solve_batched(probs, cb, pbs_, 1e-6, 1e-8, 1e-8, max_nfev=max_nfev, ctx=ctx)
        assert sum(solved_alone) <= 1      # (a problem whose reduced system needs a second damping leaves the batch: allowed)
        nf = []
        for (c, p_, rep), c2, p2, rep2 in zip(alone, cb, pbs_, reps):
            assert (rep.nfev, rep.njev, rep.status, rep.iterations) == (rep2.nfev, rep2.njev, rep2.status, rep2.iterations)
            assert rep.cost == rep2.cost and rep.cost0 == rep2.cost0 and rep.optimality == rep2.optimality
            assert torch.equal(c, c2) and torch.equal(p_, p2)
            nf.append(rep.nfev)
        if max_nfev is None:
            assert len(set(nf)) > 1 and all(r.status > 0 for r in reps)      # (the lock-step had problems finishing early)
        else:
            assert all(r.nfev <= 7 for r in reps) and any(r.status == 0 for r in reps)
    # a problem the batched kernels cannot take (10 cameras: the reduced system is smaller than two Cholesky blocks) sends
    # the call down the one-by-one road: same results, flagged
    pr = synth.make_ba_problem(10, 500, 5, seed=9)
    with np.errstate(all="ignore"):
        cams0 = bundleAdjuster.frameParameters(pr["ext"]).reshape(10, 6)
    small = ops.BADevice(pr["K"], pr["fi"], pr["pi"], pr["obs"], 10, 500, DEV, ctx)
    c_s, p_s = dev(cams0), dev(pr["pts0"].copy())
    c_a, p_a = c_s.clone(), p_s.clone()
    rep_a, _ = small.trf_solve(c_a, p_a, 1e-6, 1e-8, 1e-8)
    cb, pbs_ = [x0[0][0].clone(), c_s.clone()], [x0[0][1].clone(), p_s.clone()]
    reps, solved_alone = ops.trf_solve_batched([probs[0], small], cb, pbs_, 1e-6, 1e-8, 1e-8, ctx=ctx)
    assert solved_alone == [True, True] and ctx.control(ctx.CTL_BATCH_LAST) == 0
    assert torch.equal(cb[1], c_a) and torch.equal(pbs_[1], p_a) and reps[1].nfev == rep_a.nfev and reps[1].cost == rep_a.cost


def test_clip_pipeline_c2_shape_match_and_triangulate():
    """BASELINE config "1080p, 2000 key points, BF match + 2-view triangulation" on a short clip: the batched pipeline's
    matches of two frame pairs equal the oracle's (detect -> describe -> kNN-2 -> ratio, bit exact at 1080p), every
    track's DLT point equals the oracle's DLT of its first / last observation, and well-conditioned tracks reproject
    to their key points."""
    F = 6
    frames, ext, K = synth.render_orbit_frames_torch(F, 1920, 1080, DEV, arc_deg=0.72 * F)
    pipe = ClipPipeline(1080, 1920, 2000, batch=F)
    out = pipe.run(frames, K, ext, ba=False)
    ClipPipeline.tracks_to_host(out)
    host = frames.cpu().numpy()
    o = {i: oo.detect_compute(host[i], 2000, brief_pattern()) for i in (0, 1, 2)}
    for k in (0, 1):
        idx, dist = oo.bf_knn2(o[k]["desc"], o[k + 1]["desc"])
        want = oo.ratio_filter(idx, dist, 0.75)
        assert out["match_count"][k] == len(want)
    xy = out["xy_dev"].cpu().numpy()
    for i in (0, 1, 2):
        assert np.array_equal(xy[i, :o[i]["n"]], o[i]["xy"])
    tp, of, ok = out["track_ptr"], out["obs_frame"], out["obs_kp"]
    first, last = tp[:-1], tp[1:] - 1
    proj = np.einsum("ij,fjk->fik", np.asarray(K, float), np.asarray(ext, float)[:, :3, :])
    x0 = xy[of[first], ok[first]].astype(np.float64)
    x1 = xy[of[last], ok[last]].astype(np.float64)
    Xo = bo.triangulate_dlt(proj[of[first]], proj[of[last]], x0, x1)
    X = out["points0"].cpu().numpy()
    good = np.isfinite(Xo).all(1) & (np.abs(Xo).max(1) < 1e3) & (of[last] - of[first] >= 2)
    assert good.sum() > 200
    np.testing.assert_allclose(X[good], Xo[good], rtol=1e-6, atol=1e-7)
    # reprojection of the triangulated point into its first view (a consistency property; mismatches are excluded
    # by asking for the median)
    Xh = np.hstack([X[good], np.ones((good.sum(), 1))])
    u = np.einsum("nij,nj->ni", proj[of[first]][good], Xh)
    err = np.linalg.norm(u[:, :2] / u[:, 2:3] - x0[good], axis=1)
    assert np.median(err) < 2.0


def test_clip_pipeline_with_ba_reduces_reprojection_error():
    frames, ext, K = synth.render_orbit_frames(6, 640, 480, arc_deg=8.0)
    pipe = ClipPipeline(480, 640, 1000, batch=6)
    out = pipe.run(dev(frames), K, ext, ba=True)
    res = out["ba"]
    assert res.status in (1, 2, 3, 4) and np.isfinite(res.cost)
    assert out["n_obs"] >= 2 * out["n_tracks"] > 200


# ============================================================================================== N > 1 on one GPU

def _run_dist_workers(tmp_path, world):
    """`world` fresh child processes (never forked from this GPU-initialised one), gloo, all on cuda:0."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dist_gpu_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} of {world} failed:\n{o[-4000:]}"
    return [np.load(os.path.join(str(tmp_path), f"w{world}_rank{r}.npz")) for r in range(world)]


def test_two_ranks_on_one_gpu_real_kernels_match_one_rank(tmp_path):
    """The N > 1 path with the REAL HIP kernels under a process group: two ranks (gloo, both on
    cuda:0) against one rank.  Sharded BA: same nfev, cost within 1e-9, replicated cameras bit-identical across ranks
    — once with the packed band exchange (every shard has a pair list) and once with a shard that has NO pair list
    (a track spanning > 192 cameras): every rank must enter the same collectives.  Whole clip: identical tracks on every rank;
    sharded sliding-window BA: same window table as one rank."""
    one = _run_dist_workers(tmp_path, 1)[0]
    r0, r1 = _run_dist_workers(tmp_path, 2)
    for tag in ("band", "long"):
        assert int(r0[f"{tag}_nfev"]) == int(r1[f"{tag}_nfev"]) == int(one[f"{tag}_nfev"]), tag
        assert int(r0[f"{tag}_status"]) == int(one[f"{tag}_status"]) > 0
        c1, c2 = float(one[f"{tag}_cost"]), float(r0[f"{tag}_cost"])
        assert float(r1[f"{tag}_cost"]) == c2 and abs(c2 - c1) <= 1e-9 * c1, (tag, c1, c2)
        np.testing.assert_array_equal(r0[f"{tag}_cams"], r1[f"{tag}_cams"])
        assert int(r0[f"{tag}_hi"]) == int(r1[f"{tag}_lo"])
        pts = np.concatenate([r0[f"{tag}_pts"], r1[f"{tag}_pts"]])
        assert pts.shape == one[f"{tag}_pts"].shape
        # summation order differs between 1 and 2 ranks; on a converged, gauge-free problem x agrees to ~1e-6
        np.testing.assert_allclose(pts, one[f"{tag}_pts"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(r0[f"{tag}_cams"], one[f"{tag}_cams"], rtol=0, atol=1e-5)
        # the sharded solve is the LIBRARY loop (mm_ba_trf_dist): seven packed collectives per evaluation, three to start
        if os.environ.get("MM_TRF_DRIVER", "library") == "library":
            n_coll, nfev = int(r0[f"{tag}_collectives"]), int(r0[f"{tag}_nfev"])
            assert int(one[f"{tag}_collectives"]) == 0 and n_coll == int(r1[f"{tag}_collectives"])
            assert 0 < n_coll <= 7 * nfev + 3, (tag, n_coll, nfev)
    assert int(r0["band_n_pairs"]) > 0 and int(r1["band_n_pairs"]) > 0              # packed band exchange taken
    assert int(r0["long_n_pairs"]) > 0 and int(r1["long_n_pairs"]) == 0              # mixed shards -> dense, same everywhere
    assert int(r1["long_cam_span"]) > 192
    # clip: detection / matching sharded over the ranks, gathered on the device, linked identically everywhere
    for k in ("clip_track_ptr", "clip_obs_frame", "clip_obs_kp", "clip_match_count", "clip_kp_count", "clip_points0"):
        np.testing.assert_array_equal(r0[k], r1[k])
        np.testing.assert_array_equal(r0[k], one[k])
    np.testing.assert_array_equal(r0["clip_cams"], r1["clip_cams"])
    assert int(r0["clip_n_pairs"]) > 0
    # real matches carry outliers (no RANSAC, as the reference): the path is chaotic, the outcome is compared
    # (well-conditioned problems are held to nfev-equal / 1e-9 above; here last-bit differences of the summation order
    # decide which outlier the path chases, so only "same ballpark, same decisions on every rank" is asserted)
    assert float(r0["clip_cost"]) == float(r1["clip_cost"]) and int(r0["clip_nfev"]) == int(r1["clip_nfev"])
    assert abs(float(r0["clip_cost"]) - float(one["clip_cost"])) <= 0.3 * float(one["clip_cost"])
    # ... and ONE trust-region iteration of that same real-match problem is pinned tightly: the sharded arithmetic (partial
    # B / g_c / S / v summed over two ranks) differs from one rank's by the order of the sums only
    assert int(one["step_nfev"]) == int(r0["step_nfev"]) == int(r1["step_nfev"]) == 2
    np.testing.assert_array_equal(r0["step_cams"], r1["step_cams"])
    np.testing.assert_array_equal(r0["step_pts"], r1["step_pts"])
    assert float(r0["step_cost"]) == float(r1["step_cost"])
    assert abs(float(r0["step_cost"]) - float(one["step_cost"])) <= 1e-9 * float(one["step_cost"])
    sc = max(np.abs(one["step_cams"]).max(), 1.0)
    assert np.abs(r0["step_cams"] - one["step_cams"]).max() <= 1e-9 * sc
    sp = max(np.median(np.abs(one["step_pts"])), 1.0)
    assert np.median(np.abs(r0["step_pts"] - one["step_pts"])) <= 1e-9 * sp
    # the factorisation path is a GROUP decision: one rank told to avoid the single launch / one rank's
    # factorisation abandoned -> both ranks end on the launch-per-column path with bit-identical cameras
    if os.environ.get("MM_TRF_DRIVER", "library") == "library":
        for tag in ("avoid", "abandon"):
            np.testing.assert_array_equal(r0[f"{tag}_cams"], r1[f"{tag}_cams"])
            assert float(r0[f"{tag}_cost"]) == float(r1[f"{tag}_cost"]) and int(r0[f"{tag}_nfev"]) == int(r1[f"{tag}_nfev"])
            assert int(r0[f"{tag}_last_path"]) == int(r1[f"{tag}_last_path"]) == 0, tag
            assert int(one[f"{tag}_last_path"]) == 1
            assert abs(float(r0[f"{tag}_cost"]) - float(one[f"{tag}_cost"])) <= 1e-9 * float(one[f"{tag}_cost"])
        assert int(r0["abandon_fallbacks"]) == int(r1["abandon_fallbacks"]) == 1
    np.testing.assert_array_equal(r0["win_points"], one["win_points"])
    np.testing.assert_array_equal(r0["win_cams"], r1["win_cams"])
    np.testing.assert_array_equal(r0["win_pts"], r1["win_pts"])
    np.testing.assert_array_equal(r0["win_nfev"], r1["win_nfev"])
    # wavefront schedule: whole windows spread over the ranks -> the result does not depend on the world size, bit for bit
    for k in ("wf_cams", "wf_pts", "wf_table"):
        np.testing.assert_array_equal(r0[k], r1[k])
        np.testing.assert_array_equal(r0[k], one[k])
    assert len(one["wf_table"]) == len(one["win_points"]) and set(one["wf_table"][:, 7]) >= {0.0, 1.0}
    assert np.all(np.abs(r0["win_cost"] - one["win_cost"]) <= 0.3 * one["win_cost"])


def _run_bench(args, env_extra, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, cwd=root)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself (fresh children of a
    parent that never touches HIP) and rank 0's JSON line says n_gpus = N.  Two ranks share this one GPU through gloo -- a
    functional rehearsal of the driver's command shape, not a timing."""
    p = _run_bench(["--gpus", "2", "--frames", "48", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                   {"MM_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    if os.environ.get("MM_TRF_DRIVER", "library") == "library":
        assert line["ba"]["driver"] == "mm_ba_trf_dist" and line["ba"]["collectives"] > 0
    assert line["value"] > 0 and line["problem"]["tracks"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    """... and `--gpus 8` on a box with fewer GPUs fails loudly instead of reporting a one-GPU run as eight."""
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    p = _run_bench(["--gpus", "8", "--frames", "16", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {}, timeout=120)
    assert p.returncode != 0
    assert b"GPU(s) visible" in p.stderr and not any(ln.startswith(b"{") for ln in p.stdout.splitlines())


def test_library_dist_loop_over_rccl_one_rank(tmp_path):
    """backend "nccl" (RCCL) under the sharded library loop, as far as one GPU allows: one rank whose all-reduce always goes
    through torch.distributed.  The sums are trivial, the path is the real one (ProcessGroupNCCL on views of the library's
    workspace, ordered on the library's stream, seven calls per evaluation): same evaluations and cost as mm_ba_trf."""
    _needs_library_driver()
    import socket
    import subprocess
    import sys
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    out = os.path.join(str(tmp_path), "rccl1.npz")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_rccl_one_rank_worker.py")
    p = subprocess.run([sys.executable, worker, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    d = np.load(out)
    assert str(d["backend"]) == "nccl"
    assert int(d["nfev"][0]) == int(d["nfev"][1]) and int(d["status"][0]) == int(d["status"][1]) > 0
    assert abs(float(d["cost"][0]) - float(d["cost"][1])) <= 1e-10 * float(d["cost"][0])
    np.testing.assert_allclose(d["cams_rccl"], d["cams_plain"], rtol=0, atol=1e-7)
    # every collective of the library loop went through the process group (+ the initial cost and the span reduction of the
    # Python prologue)
    assert int(d["collectives"]) > 0 and int(d["calls"]) == int(d["collectives"]) + 2
    assert int(d["collectives"]) <= 7 * int(d["nfev"][1]) + 3
    # the clip path through the same backend: four device all-gathers, identical tracks, the same adjustment
    assert int(d["clip_gathers"]) == 4 and bool(d["clip_tracks_equal"])
    assert int(d["clip_nfev"][0]) == int(d["clip_nfev"][1]) and int(d["clip_collectives"]) > 0
    assert abs(float(d["clip_cost"][0]) - float(d["clip_cost"][1])) <= 1e-9 * float(d["clip_cost"][0])


# ============================================================================================== keyframe gating, contrast

from oracle import frame_oracle as fo  # noqa: E402
from meatmodeler_amd import frame_tables  # noqa: E402


@pytest.mark.parametrize("w,h", [(640, 480), (333, 201), (65, 17)])
def test_pyr_down_bit_exact(w, h):
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    lv = ops.pyramid(dev(img), 3)
    ref = fo.pyramid(img, 3)
    for a, b in zip(lv, ref):
        np.testing.assert_array_equal(a.cpu().numpy(), b)


@pytest.mark.parametrize("win,levels,count", [((21, 21), 3, 30), ((15, 15), 2, 10), ((9, 13), 0, 5), ((41, 41), 4, 30)])
def test_lk_track_bit_exact(win, levels, count):
    """calcOpticalFlowPyrLK (processor.py:79): next points, status and error bit for bit against the integer-exact CPU
    definition, on two consecutive rendered frames with corners from goodFeaturesToTrack plus points at / beyond the
    border (status 0 paths)."""
    frames, _, _ = synth.render_orbit_frames(2, 640, 480, arc_deg=1.5, seed=4)
    pts = fo.good_features(frames[0], 300, 0.02, 7, 5)
    pts = np.vstack([pts, [[0.4, 0.2], [639.0, 479.0], [-30.0, 50.0], [320.5, 240.25], [700.0, 10.0]]]).astype(np.float32)
    nx_o, st_o, er_o = fo.lk_track(frames[0], frames[1], pts, win, levels, count, 0.03)
    pp, pn = ops.pyramid(dev(frames[0]), levels), ops.pyramid(dev(frames[1]), levels)
    nx, st, er = ops.lk_track(pp, pn, dev(pts), win, count, 0.03)
    np.testing.assert_array_equal(st.cpu().numpy(), st_o)
    np.testing.assert_array_equal(nx.cpu().numpy(), nx_o)
    np.testing.assert_array_equal(er.cpu().numpy(), er_o)
    assert st_o.sum() > 200 and st_o[-1] == 0 and st_o[-3] == 0


def test_calc_optical_flow_without_points_fails_like_cv2():
    """prevPts = None (goodFeaturesToTrack found nothing at the last keyframe): the cv2 call of the reference raises
    (processor.py:79); an empty point array yields empty outputs.  And the pyramid cache only trusts frozen arrays that own
    their data: a read-only VIEW of a writable frame is re-uploaded when the frame changes underneath."""
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    b = np.roll(a, 2, axis=1)
    with pytest.raises(ValueError):
        processor.calcOpticalFlowPyrLK(a, b, None)
    assert processor.calcOpticalFlowPyrLK(a, b, np.zeros((0, 1, 2), np.float32)) == (None, None, None)
    pts = np.array([[[80.0, 60.0]]], np.float32)
    view = a.view()
    view.flags.writeable = False
    p1, st1, _ = processor.calcOpticalFlowPyrLK(view, b, pts)
    a[:] = np.roll(a, 5, axis=1)                     # the base changes behind the read-only view
    p2, st2, _ = processor.calcOpticalFlowPyrLK(view, b, pts)
    po, so, _ = fo.lk_track(a, b, pts.reshape(-1, 2), (21, 21), 3, 30, 0.01)
    np.testing.assert_array_equal(p2.reshape(-1, 2), po)
    assert st1[0, 0] == 1 and not np.array_equal(p1, p2)


def test_lk_track_recovers_a_known_shift():
    frames, _, _ = synth.render_orbit_frames(1, 640, 480, arc_deg=1.0, seed=2)
    a = frames[0]
    b = np.roll(np.roll(a, 2, axis=0), 3, axis=1)
    pts = fo.good_features(a, 100, 0.05, 15, 5)
    pts = pts[(pts[:, 0] > 40) & (pts[:, 0] < 600) & (pts[:, 1] > 40) & (pts[:, 1] < 440)]
    p, st, err = processor.calcOpticalFlowPyrLK(a, b, pts.reshape(-1, 1, 2), None, winSize=(21, 21), maxLevel=3,
                                                criteria=(3, 30, 0.01))
    assert p.shape == (len(pts), 1, 2) and st.shape == (len(pts), 1) and err.dtype == np.float32
    d = (p[:, 0] - pts)[st[:, 0] == 1]
    assert st.mean() > 0.95 and np.abs(np.median(d, 0) - [3, 2]).max() < 0.05


@pytest.mark.parametrize("bs,w,h", [(3, 640, 480), (7, 333, 201), (5, 64, 40)])
def test_min_eig_and_good_features_bit_exact(bs, w, h):
    """goodFeaturesToTrack (processor.py:104): eigenvalue map bit for bit, same corners in the same order."""
    frames, _, _ = synth.render_orbit_frames(1, w, h, arc_deg=1.0, seed=bs)
    img = frames[0]
    np.testing.assert_array_equal(ops.min_eig(dev(img), bs).cpu().numpy(), fo.min_eig(img, bs))
    for max_c, q, md in ((200, 0.01, 10.0), (0, 0.1, 3.0), (50, 0.3, 0.0), (1000, 0.001, 7.5)):
        got = ops.good_features(dev(img), max_c, q, md, bs)
        ref = fo.good_features(img, max_c, q, md, bs)
        np.testing.assert_array_equal(got, ref, err_msg=str((max_c, q, md)))
    assert len(fo.good_features(img, 200, 0.01, 10.0, bs)) > (20 if w > 100 else 3)
    assert processor.goodFeaturesToTrack(np.full((40, 40), 7, np.uint8), 10, 0.1, 5) is None      # flat image: no corners
    # a periodic pattern: hundreds of corners with EQUAL strength and plateaus of equal values -> the tie order (y, x)
    # and the candidate capacity are exercised
    yy, xx = np.mgrid[0:h, 0:w]
    board = (((yy // 8) + (xx // 8)) % 2 * 200 + 20).astype(np.uint8)
    np.testing.assert_array_equal(ops.min_eig(dev(board), bs).cpu().numpy(), fo.min_eig(board, bs))
    for max_c, q, md in ((0, 0.5, 0.0), (300, 0.2, 6.0)):
        np.testing.assert_array_equal(ops.good_features(dev(board), max_c, q, md, bs), fo.good_features(board, max_c, q, md, bs))


def test_keyframe_tracking_flow_matches_oracle_flow():
    """keyframeTracking (processor.py:61-110) over a short clip, call by call against the same function built from the
    oracle's LK / GFTT: same keyframe decisions, same points, same accumulated error."""
    frames, _, _ = synth.render_orbit_frames(8, 640, 480, arc_deg=14.0, seed=3)
    lk = dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
    fp = dict(maxCorners=100, qualityLevel=0.3, minDistance=7, blockSize=7)

    def oracle_kt(frame, prev, pts, acc, thr):
        if pts is None or len(pts) == 0:
            return False, prev, pts, acc
        nx, st, er = fo.lk_track(prev, frame, pts.reshape(-1, 2), lk["winSize"], lk["maxLevel"], 10, 0.03)
        pts = nx[st == 1].reshape(-1, 1, 2)
        e = np.nan_to_num(er.reshape(-1, 1))
        e[e < 0] = 0
        acc += np.average(e)
        if acc > thr * frame.shape[1]:
            c = fo.good_features(frame, fp["maxCorners"], fp["qualityLevel"], fp["minDistance"], fp["blockSize"])
            return True, frame, (c.reshape(-1, 1, 2) if len(c) else None), 0
        return False, frame, pts, acc

    p0 = processor.goodFeaturesToTrack(frames[0], mask=None, **fp)
    np.testing.assert_array_equal(p0[:, 0], fo.good_features(frames[0], 100, 0.3, 7, 7))
    prev, pts, acc = frames[0], p0, 0.0
    oprev, opts, oacc = frames[0], p0.copy(), 0.0
    keys = []
    for k in range(1, 8):
        is_k, prev, pts, acc = processor.keyframeTracking(frames[k], prev, pts, acc, lk, fp, threshold=0.03)
        ois, oprev, opts, oacc = oracle_kt(frames[k], oprev, opts, oacc, 0.03)
        assert is_k == ois and acc == oacc
        np.testing.assert_array_equal(pts, opts)
        assert np.shares_memory(prev, frames[k])
        keys.append(is_k)
    assert any(keys) and not all(keys)


@pytest.mark.parametrize("w,h", [(640, 480), (333, 203)])
def test_increase_contrast_and_grey_bit_exact(w, h):
    """increaseContrast + COLOR_BGR2GRAY (processor.py:12-26, :357) against the fixed-point CPU definition, incl. a size
    that the 8 x 8 tile grid does not divide (reflected padding)."""
    frames, _, _ = synth.render_orbit_frames(2, w, h, arc_deg=3.0, seed=8)
    rng = np.random.default_rng(1)
    bgr = np.stack([frames[0], np.roll(frames[0], 7, 1), (frames[1] * 0.6).astype(np.uint8) + rng.integers(0, 40, (h, w), dtype=np.uint8)], -1)
    ref = fo.increase_contrast(bgr, frame_tables.lab_tables())
    got = processor.increaseContrast(bgr)
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(processor.cvtColorBGR2GRAY(got), fo.bgr_to_grey(ref))
    out, grey = ops.increase_contrast(dev(np.stack([bgr, bgr[::-1].copy()])), want_grey=True)      # batched + fused grey
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref)
    np.testing.assert_array_equal(grey[0].cpu().numpy(), fo.bgr_to_grey(ref))
    np.testing.assert_array_equal(out[1].cpu().numpy(), fo.increase_contrast(bgr[::-1].copy(), frame_tables.lab_tables()))
    assert ref.std() > bgr.std()          # it does increase the contrast


def test_process_frames_driver_loop_matches_oracle_flow(tmp_path):
    """processor.processFrames = the body of the reference's `process` (processor.py:356-485) on decoded BGR frames:
    contrast, grey, keyframe gate, ORB + matching + track linking on keyframes, triangulation, flattening, bundle
    adjustment, PLY.  Same keyframes and the same flattened observations as the loop rebuilt from the oracle's functions."""
    frames_g, ext, K = synth.render_orbit_frames(10, 320, 240, arc_deg=18.0, seed=6)
    frames = [np.stack([f, np.roll(f, 3, 1), np.roll(f, 2, 0)], -1) for f in frames_g]
    lk = dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
    fp = dict(maxCorners=80, qualityLevel=0.2, minDistance=7, blockSize=7)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = processor.processFrames(frames, K, lambda i: ext[i], str(tmp_path) + "/run", lk, fp, threshold=0.06, nfeatures=500)
    # ---- the same loop from the oracle's functions ----
    tabs = frame_tables.lab_tables()
    grey = [fo.bgr_to_grey(fo.increase_contrast(f, tabs)) for f in frames]
    pts = fo.good_features(grey[0], 80, 0.2, 7, 7).reshape(-1, 1, 2)
    prev, acc, keys = grey[0], 0.0, [0]
    o_prev = oo.detect_compute(grey[0], 500, brief_pattern())
    tracks, popped, pk, k = [], [], 0, 1
    for i in range(1, 10):
        nx, st, er = fo.lk_track(prev, grey[i], pts.reshape(-1, 2), (15, 15), 2, 10, 0.03)
        pts = nx[st == 1].reshape(-1, 1, 2)
        prev = grey[i]
        e = np.nan_to_num(er.reshape(-1, 1))
        e[e < 0] = 0
        acc += np.average(e)
        if acc > 0.06 * 320:
            acc = 0
            c = fo.good_features(grey[i], 80, 0.2, 7, 7)
            pts = c.reshape(-1, 1, 2)
            keys.append(i)
            o_new = oo.detect_compute(grey[i], 500, brief_pattern())
            io_, do_ = oo.bf_knn2(o_prev["desc"], o_new["desc"])
            good = oo.ratio_filter(io_, do_, 0.75)
            p, tracks = bo.point_tracking(tracks, pk, o_prev["xy"][good[:, 0]].astype(np.float64), k,
                                          o_new["xy"][good[:, 1]].astype(np.float64))
            popped += p
            o_prev, pk, k = o_new, k, k + 1
    popped += tracks
    assert out["keyframes"] == keys and 2 < len(keys) < 10
    assert len(out["tracks"]) == len(popped) > 20
    for t, ot in zip(out["tracks"], popped):
        assert list(t.getCoordinates().items()) == list(ot.getCoordinates().items())
    assert out["points"].shape == (len(popped), 3) and len(out["extrinsics"]) == len(keys)
    assert os.path.exists(out["file"]) and out["file"].endswith("runCloud.ply")
    assert "termination condition is satisfied" in buf.getvalue() or "maximum number" in buf.getvalue()
