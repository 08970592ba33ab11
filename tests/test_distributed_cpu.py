"""CPU, world_size = 2, gloo: the N > 1 path.

* frame-pair / point partitions and the variable-length all-gather that rebuilds identical match lists on every rank;
* the sharded trust-region driver (`SchurTRF` with an `AllReduce`): points are split over the ranks, cameras are
  replicated, camera-side blocks / the reduced camera system / a few scalars are all-reduced.  The HIP sweeps cannot
  run here, so the test plugs a NumPy stand-in for `ops.BADevice` (built on the oracle's cost function) into the
  product driver — what is under test is the driver's sharding algebra and collective placement, which must
  reproduce the single-process iterates.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from meatmodeler_amd import parallel, synth  # noqa: E402
from oracle import ba_oracle as bo  # noqa: E402


class NumpyBA:
    """Test double for ops.BADevice on CPU tensors (dense NumPy algebra; central-difference Jacobian blocks)."""

    def __init__(self, K, fi, pi, obs, F, P):
        self.K, self.fi, self.pi, self.obs = np.asarray(K, float), np.asarray(fi), np.asarray(pi), np.asarray(obs, float)
        self.F, self.P, self.O = F, P, len(fi)
        self.device = torch.device("cpu")
        self.ctx = None
        self.n_pairs = 0
        span = 0
        for p in range(P):
            f = self.fi[self.pi == p]
            if f.size:
                span = max(span, int(f.max() - f.min()))
        self.cam_span = span

    def _x(self, cams, pts):
        return np.hstack([cams.numpy().ravel(), pts.numpy().ravel()])

    def _res(self, cams, pts):
        return bo.point_fun(self._x(cams, pts), self.K, self.F, self.P, self.fi, self.pi, self.obs).reshape(-1, 2)

    def _jac(self, cams, pts):
        return bo.jacobian_fd(self._x(cams, pts), self.K, self.F, self.P, self.fi, self.pi, self.obs, h=1e-6)

    def residual(self, cams, pts, want_res=False):
        r = self._res(cams, pts)
        return torch.tensor([float((r * r).sum())], dtype=torch.float64), (torch.from_numpy(r) if want_res else None)

    def normal_eq(self, cams, pts, want_cams=True, want_pts=True):
        Jc, Jp = self._jac(cams, pts)
        r = self._res(cams, pts)
        B = np.zeros((self.F, 6, 6)); gc = np.zeros((self.F, 6)); C = np.zeros((self.P, 3, 3)); gp = np.zeros((self.P, 3))
        np.add.at(B, self.fi, np.einsum("omi,omj->oij", Jc, Jc))
        np.add.at(gc, self.fi, np.einsum("omi,om->oi", Jc, r))
        np.add.at(C, self.pi, np.einsum("omi,omj->oij", Jp, Jp))
        np.add.at(gp, self.pi, np.einsum("omi,om->oi", Jp, r))
        C6 = np.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1)
        return torch.from_numpy(B), torch.from_numpy(gc), torch.from_numpy(C6), torch.from_numpy(gp)

    def jvp(self, cams, pts, wc, wp):
        Jc, Jp = self._jac(cams, pts)
        out = np.einsum("omi,oi->om", Jc, wc.numpy()[self.fi]) + np.einsum("omi,oi->om", Jp, wp.numpy()[self.pi])
        return torch.from_numpy(out)

    def multi_dot(self, pairs, split=0):
        rows = [[torch.dot(a[:split], b[:split]), torch.dot(a[split:], b[split:]), torch.dot(a, b)] for a, b in pairs]
        return torch.stack([torch.stack(r) for r in rows])

    def trf_fused(self, op, ins, outs, scalars=(), h0=0.0, h1=0.0, split=0):
        """torch restatement of mm_trf_fused (see include/meatmodeler.h)."""
        def rows(pairs, mx=None):
            r = [[torch.dot(a[:split], b[:split]), torch.dot(a[split:], b[split:]), torch.dot(a, b)] for a, b in pairs]
            if mx is None:
                r.append([torch.zeros((), dtype=torch.float64)] * 3)
            else:
                m0 = mx[:split].abs().max() if split else torch.zeros((), dtype=torch.float64)
                m1 = mx[split:].abs().max() if split < mx.numel() else torch.zeros((), dtype=torch.float64)
                r.append([m0, m1, torch.maximum(m0, m1)])
            return torch.stack([torch.stack(list(x)) for x in r])
        if op == 0:
            g, si = ins
            outs[0].copy_(g / si)
            outs[1].copy_(outs[0] / si)
            return rows([(outs[0], outs[0])], g)
        if op == 1:
            v, dp, si, gh = ins
            outs[0].copy_(torch.cat([v, dp]) * si)
            outs[1].copy_(gh / torch.sqrt(scalars[0][0]))
            return rows([(outs[1], outs[0]), (outs[0], outs[0])])
        if op == 2:
            gn, q1 = ins
            outs[0].copy_(gn - scalars[0][0] * q1)
            return rows([(outs[0], outs[0])])
        if op == 3:
            w, q1, si, gh, x = ins
            outs[0].copy_(w / torch.sqrt(scalars[0][0]))
            outs[1].copy_(q1 / si)
            outs[2].copy_(outs[0] / si)
            return rows([(outs[1], outs[1]), (outs[1], outs[2]), (outs[2], outs[2]), (outs[0], gh), (x, x)])
        x, s1, s2 = ins
        outs[0].copy_(x + h0 * s1 + h1 * s2)
        return torch.zeros((1, 3), dtype=torch.float64)

    def trf_damping(self, gh2, d11, Delta, min_damping):
        """SciPy trf.py:473-477 (the product runs this as a one-thread kernel, mm_trf_damping)."""
        a, b = 0.5 * float(d11), -float(gh2)
        to_tr = Delta / np.sqrt(float(gh2))
        ts = [0.0, to_tr]
        if a != 0 and 0.0 < -0.5 * b / a < to_tr:
            ts.append(-0.5 * b / a)
        reg = -min(t * (a * t + b) for t in ts) / Delta ** 2
        return torch.tensor([reg, max(reg, min_damping)], dtype=torch.float64)

    @staticmethod
    def _sym(C6):
        C = np.zeros((len(C6), 3, 3))
        idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
        for k, (i, j) in enumerate(idx):
            C[:, i, j] = C6[:, k]
            C[:, j, i] = C6[:, k]
        return C

    def schur(self, cams, pts, Bd, Cd, gc, gp):
        Jc, Jp = self._jac(cams, pts)
        Ci = np.linalg.inv(self._sym(Cd.numpy()))
        n = 6 * self.F
        S = np.zeros((n, n))
        for f in range(self.F):
            S[6 * f:6 * f + 6, 6 * f:6 * f + 6] = Bd.numpy()[f]
        v = gc.numpy().ravel().copy()
        E = np.einsum("omi,omj->oij", Jc, Jp)                       # [O,6,3]
        for p in range(self.P):
            obs_p = np.flatnonzero(self.pi == p)
            for o in obs_p:
                Y = E[o] @ Ci[p]
                v[6 * self.fi[o]:6 * self.fi[o] + 6] -= Y @ gp.numpy()[p]
                for o2 in obs_p:
                    S[6 * self.fi[o]:6 * self.fi[o] + 6, 6 * self.fi[o2]:6 * self.fi[o2] + 6] -= Y @ E[o2].T
        Ci6 = np.stack([Ci[:, 0, 0], Ci[:, 0, 1], Ci[:, 0, 2], Ci[:, 1, 1], Ci[:, 1, 2], Ci[:, 2, 2]], 1)
        return torch.from_numpy(S), torch.from_numpy(v), torch.from_numpy(Ci6)

    def chol_solve(self, S, v, half_bandwidth=None):
        A = S.numpy()
        A = np.tril(A) + np.tril(A, -1).T
        try:
            np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            return torch.tensor([1], dtype=torch.int32)
        v.copy_(torch.from_numpy(np.linalg.solve(A, v.numpy())))
        return torch.tensor([0], dtype=torch.int32)

    def backsub(self, cams, pts, Cinv, gp, dc):
        Jc, Jp = self._jac(cams, pts)
        t = gp.numpy().copy()
        s = np.einsum("omi,oi->om", Jc, dc.numpy()[self.fi])
        np.subtract.at(t, self.pi, np.einsum("omi,om->oi", Jp, s))
        Ci = self._sym(Cinv.numpy())
        return torch.from_numpy(np.einsum("pij,pj->pi", Ci, t))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from meatmodeler_amd.bundleAdjuster import SchurTRF
        # --- variable-length gather of per-rank match lists ---
        F = 9
        (p_lo, p_hi), (f_lo, f_hi) = parallel.pair_block(F, rank, world)
        local = np.arange(p_lo * 10, p_hi * 10, dtype=np.int32)           # 10 numbers per owned pair
        parts = parallel.gather_varlen(local, world, dist)
        assert np.array_equal(np.concatenate(parts), np.arange(0, (F - 1) * 10, dtype=np.int32))
        assert f_hi - f_lo == (p_hi - p_lo) + 1                             # halo frame
        # --- sharded BA ---
        Fc, P, L = 6, 40, 4
        pr = synth.make_ba_problem(Fc, P, L, seed=3)
        cams0 = torch.from_numpy(bo.frame_parameters(pr["ext"]).reshape(Fc, 6))
        lo, hi, mask = parallel.partition_points(pr["fi"], pr["pi"], P, rank, world)
        pb = NumpyBA(pr["K"], pr["fi"][mask], pr["pi"][mask] - lo, pr["obs"][mask], Fc, hi - lo)
        solver = SchurTRF(pb, allreduce=parallel.AllReduce())
        res = solver.solve(cams0, torch.from_numpy(pr["pts0"][lo:hi].copy()), ftol=1e-4, verbose=0)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), cams=res.cams.numpy(), pts=res.pts.numpy(), lo=lo, hi=hi,
                 cost=res.cost, nfev=res.nfev, status=res.status)
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_sharded_ba_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # single process, same driver, same stand-in
    from meatmodeler_amd.bundleAdjuster import SchurTRF
    Fc, P, L = 6, 40, 4
    pr = synth.make_ba_problem(Fc, P, L, seed=3)
    pb = NumpyBA(pr["K"], pr["fi"], pr["pi"], pr["obs"], Fc, P)
    res = SchurTRF(pb).solve(torch.from_numpy(bo.frame_parameters(pr["ext"]).reshape(Fc, 6)),
                             torch.from_numpy(pr["pts0"].copy()), ftol=1e-4, verbose=0)
    assert int(r0["nfev"]) == int(r1["nfev"]) == res.nfev and int(r0["status"]) == res.status
    assert abs(float(r0["cost"]) - res.cost) <= 1e-9 * res.cost and float(r0["cost"]) == float(r1["cost"])
    np.testing.assert_allclose(r0["cams"], r1["cams"], rtol=0, atol=0)            # replicated cameras stay identical
    # Summation order differs between 1 and 2 ranks; on this gauge-free, weakly constrained problem (F=6, P=40) last-bit
    # differences grow to ~1e-4 in x along the gauge directions (the reference's own SciPy run shows the same
    # sensitivity, tests/test_oracle_golden.py::test_g5) while the cost agrees to 1e-9.
    np.testing.assert_allclose(r0["cams"], res.cams.numpy(), rtol=0, atol=5e-3)
    pts = np.concatenate([r0["pts"], r1["pts"]])
    assert int(r0["lo"]) == 0 and int(r0["hi"]) == int(r1["lo"]) and int(r1["hi"]) == P
    np.testing.assert_allclose(pts, res.pts.numpy(), rtol=0, atol=5e-3)
    # and the result is what the reference's SciPy recipe finds (cost within 1e-4, the north-star tolerance)
    _, _, ref = bo.adjust_points(pr["ext"], pr["K"], pr["pts0"][:, None, :], pr["obs"], pr["fi"], pr["pi"],
                                 return_result=True)
    assert res.cost <= ref.cost * (1 + 1e-4)
