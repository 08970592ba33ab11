"""Drop-in for the reference's ``bundleAdjuster.py`` — same public names, argument and return conventions.

The reference hands its cost function to ``scipy.optimize.least_squares`` (TRF, ``jac_sparsity``,
``x_scale='jac'``, ``ftol=1e-4``; bundleAdjuster.py:180-192).  SciPy then differentiates numerically and solves
each damped Gauss-Newton system with LSMR (scipy/optimize/_lsq/trf.py:401-560).  Here the same trust-region
iteration (`trf_no_bounds`: Jacobian column scaling with running max, Cauchy-derived regulariser, 2-D subspace
trust-region step, ratio test, ftol/xtol/gtol termination) runs with

  * the analytic Jacobian (never materialised) and block normal equations computed by HIP sweeps,
  * the regularised Gauss-Newton system  (J^T J + reg D^-2) q = J^T f  solved EXACTLY through the Schur complement
    onto the cameras (point blocks eliminated, dense 6F x 6F Cholesky on f64 MFMA) instead of iteratively by LSMR.

So the iterates follow SciPy's to within LSMR's own 1e-6 tolerance and the finite-difference error of the reference's
Jacobian; see DESIGN.md §BA for what "parity" means on this gauge-free problem.

Host Python only sequences launches and does the 2x2 / scalar algebra of the trust region; torch is used for
device buffers and a handful of axpy / dot reductions on the parameter vector.
"""
import os
import time
import warnings

import numpy as np
import torch
from numpy.linalg import norm as _norm
from scipy.linalg import cho_factor, cho_solve, LinAlgError
from scipy.sparse import csr_matrix

from . import ops
from ._lib import default_context, MMError

EPS = np.finfo(float).eps


# ----------------------------------------------------------------------------------------------- small helpers

def _rodrigues_matrix(rvec):
    """Axis-angle -> R, the closed form cv2.Rodrigues evaluates at bundleAdjuster.py:153,201."""
    r = np.asarray(rvec, float).reshape(3)
    th2 = float(r @ r)
    th = np.sqrt(th2)
    Kx = np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])
    if th < 1e-8:
        a, b = 1.0 - th2 / 6.0, 0.5 - th2 / 24.0
    else:
        a, b = np.sin(th) / th, (1.0 - np.cos(th)) / th2
    return np.eye(3) + a * Kx + b * (Kx @ Kx)


def frameParameters(frame_extrinsic_matrices):
    """[F,3|4,4] -> [6F] rows (rvec, tvec) (bundleAdjuster.py:105-134): theta = arccos((tr R - 1)/2) without
    clipping, axis = skew part / (2 sin theta) with 0/0 -> 0, r = axis * theta.  O(F) host glue."""
    E = np.asarray(frame_extrinsic_matrices, float)
    R = E[:, :3, :3]
    with np.errstate(invalid="ignore", divide="ignore"):
        theta = np.arccos((np.trace(R, axis1=1, axis2=2) - 1.0) * 0.5)
        two_s = 2.0 * np.sin(theta)
        axis = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1) / two_s[:, None]
        rvec = np.nan_to_num(axis) * theta[:, None]
    return np.concatenate([rvec, E[:, :3, 3]], axis=1).reshape(-1)


def pointAdjustmentSparsity(n_frames, n_points, frame_indices, point_indices):
    """Jacobian sparsity structure (bundleAdjuster.py:55-78) as CSR: observation i owns rows 2i, 2i+1 with ones in its
    camera's 6 and its point's 3 columns.  The HIP path never needs it (the block layout is implicit); provided for
    callers that used it."""
    fi = np.asarray(frame_indices).astype(np.int64)
    pi = np.asarray(point_indices).astype(np.int64)
    O = fi.size
    cols = np.empty((O, 2, 9), np.int64)
    cols[:, :, :6] = (6 * fi)[:, None, None] + np.arange(6)
    cols[:, :, 6:] = (6 * n_frames + 3 * pi)[:, None, None] + np.arange(3)
    return csr_matrix((np.ones(18 * O, dtype=int), cols.reshape(-1), np.arange(0, 18 * O + 1, 9)),
                      shape=(2 * O, 6 * n_frames + 3 * n_points))


def _dev(x, device):
    return torch.as_tensor(np.ascontiguousarray(x, np.float64)).to(device)


def project(points, frame_params, camera_matrix):
    """points [n,3], frame_params [n,6] -> pixels [n,2] on the device (bundleAdjuster.py:31-52): the residual sweep
    with zero observations and identity index maps."""
    ctx = default_context()
    n = len(points)
    ar = np.arange(n, dtype=np.int32)
    pb = ops.BADevice(camera_matrix, ar, ar, np.zeros((n, 2)), n, n, ctx.device, ctx)
    _, res = pb.residual(_dev(np.asarray(frame_params, float)[:, :6], ctx.device), _dev(points, ctx.device), True)
    return res.cpu().numpy()


def rotate(points, rot_vecs):
    """Rodrigues rotation of points by per-row axis-angle vectors (bundleAdjuster.py:7-28).  Not on the hot path (the
    sweeps fuse it); evaluated through `project` with K = I after translating by +t0 to keep z away from zero is NOT
    possible in general, so this helper uses device tensor algebra."""
    ctx = default_context()
    X = _dev(points, ctx.device)
    r = _dev(rot_vecs, ctx.device)
    th = r.norm(dim=1, keepdim=True)
    k = torch.where(th > 0, r / torch.where(th > 0, th, torch.ones_like(th)), torch.zeros_like(r))
    c, s = torch.cos(th), torch.sin(th)
    out = c * X + s * torch.cross(k, X, dim=1) + (k * X).sum(1, keepdim=True) * (1 - c) * k
    return out.cpu().numpy()


def pointFun(parameters, camera_matrix, n_frames, n_points, frame_indices, point_indices, points_2D):
    """Residual vector [2*O] (bundleAdjuster.py:81-102) from the HIP residual sweep."""
    ctx = default_context()
    parameters = np.asarray(parameters, float)
    pb = ops.BADevice(camera_matrix, frame_indices, point_indices, points_2D, n_frames, n_points, ctx.device, ctx)
    cams = _dev(parameters[:6 * n_frames].reshape(n_frames, 6), ctx.device)
    pts = _dev(parameters[6 * n_frames:].reshape(n_points, 3), ctx.device)
    return pb.residual(cams, pts, True)[1].cpu().numpy().ravel()


def poseFun(parameters, camera_intrinsic_matrix, n_frames, frame_indices, point_indices, points_3D, points_2D):
    """Pose-only residuals (bundleAdjuster.py:206-211)."""
    ctx = default_context()
    pb = ops.BADevice(camera_intrinsic_matrix, frame_indices, point_indices, points_2D, n_frames, len(points_3D),
                      ctx.device, ctx)
    cams = _dev(np.asarray(parameters, float).reshape(n_frames, 6), ctx.device)
    return pb.residual(cams, _dev(points_3D, ctx.device), True)[1].cpu().numpy().ravel()


# ----------------------------------------------------------------------------------------------- trust region (host scalars)

_COMPANION4 = np.diag(np.ones(3), -1)


def _solve_trust_region_2d(B, g, Delta):
    """2-D trust-region subproblem exactly as scipy/optimize/_lsq/common.py:171-219 solves it: the Cholesky attempt
    goes through the same LAPACK routines, the boundary case through the same quartic in t = tan(theta / 2).  The
    quartic's roots are np.roots's own recipe (eigenvalues of the companion matrix, same LAPACK call) applied directly
    when no coefficient vanishes, and the handful of candidate points is evaluated with scalar arithmetic instead of
    NumPy temporaries: this sits between two GPU launches on the host (150 us of an iteration before, ~110 after)."""
    try:
        R, lower = cho_factor(B)
        p = -cho_solve((R, lower), g)
        if np.dot(p, p) <= Delta ** 2:
            return p, True
    except (LinAlgError, ValueError):
        pass
    b00, b01, b11 = float(B[0, 0]), float(B[0, 1]), float(B[1, 1])
    g0, g1 = float(g[0]), float(g[1])
    a = b00 * Delta ** 2
    b = b01 * Delta ** 2
    c = b11 * Delta ** 2
    d = g0 * Delta
    f = g1 * Delta
    coeffs = np.array([-b + d, 2 * (a - c + f), 6 * b, 2 * (-a + c + f), -b - d])
    if coeffs[0] != 0.0 and coeffs[-1] != 0.0:
        A = _COMPANION4.copy()
        A[0, :] = -coeffs[1:] / coeffs[0]
        t = np.linalg.eigvals(A)
    else:
        t = np.roots(coeffs)
    best, best_val = None, np.inf
    for ti in t[np.isreal(t)].real.tolist():      # (in the order np.argmin would scan them: first minimum wins)
        q = 1 + ti * ti
        p0 = Delta * (2 * ti / q)
        p1 = Delta * ((1 - ti * ti) / q)
        val = 0.5 * (p0 * (b00 * p0 + b01 * p1) + p1 * (b01 * p0 + b11 * p1)) + (g0 * p0 + g1 * p1)
        if val < best_val:
            best, best_val = (p0, p1), val
    return np.array(best), False


def _update_tr_radius(Delta, actual, predicted, step_norm, bound_hit):
    if predicted > 0:
        ratio = actual / predicted
    elif predicted == actual == 0:
        ratio = 1
    else:
        ratio = 0
    if ratio < 0.25:
        Delta = 0.25 * step_norm
    elif ratio > 0.75 and bound_hit:
        Delta *= 2.0
    return Delta, ratio


def _check_termination(dF, F, dx_norm, x_norm, ratio, ftol, xtol):
    ftol_ok = dF < ftol * F and ratio > 0.25
    xtol_ok = dx_norm < xtol * (xtol + x_norm)
    if ftol_ok and xtol_ok:
        return 4
    if ftol_ok:
        return 2
    if xtol_ok:
        return 3
    return None


_MESSAGES = {-1: "Improper input parameters status returned from `leastsq`",
             0: "The maximum number of function evaluations is exceeded.",
             1: "`gtol` termination condition is satisfied.", 2: "`ftol` termination condition is satisfied.",
             3: "`xtol` termination condition is satisfied.",
             4: "Both `ftol` and `xtol` termination conditions are satisfied."}


def _print_header():
    print("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format("Iteration", "Total nfev", "Cost", "Cost reduction",
                                                          "Step norm", "Optimality"))


def _print_iteration(it, nfev, cost, red, step, opt):
    red = " " * 15 if red is None else f"{red:^15.2e}"
    step = " " * 15 if step is None else f"{step:^15.2e}"
    print(f"{it:^15}{nfev:^15}{cost:^15.4e}{red}{step}{opt:^15.2e}")


class BAResult:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class SchurTRF:
    """SciPy's trf_no_bounds (tr_solver='lsmr', x_scale='jac', linear loss) with the Gauss-Newton system solved by
    the Schur complement.  `allreduce` (optional) sums partial block quantities across ranks for the sharded path
    (points partitioned over GPUs, cameras replicated): it is called on every tensor that is a sum over
    observations.

    The parameter vector lives in ONE flat device buffer x = [cams (6F) | points (3P)] (views are handed to the sweeps),
    so every axpy / dot of the iteration is a single launch, and the host reads device scalars three times per
    iteration (before the damping is known, after the step basis is built, after each trial step)."""

    def __init__(self, pb, allreduce=None, timers=None, min_damping=1e-9, driver=None):
        self.pb = pb
        self.allreduce = allreduce
        self.timers = timers
        self.min_damping = min_damping
        # "library": the loop itself runs inside the C-ABI library (mm_ba_trf; one GPU only); "python": sequenced from
        # here (what the sharded path and the CPU stand-in of the tests use).  Same kernels, bit-identical iterates.
        self.driver = driver or os.environ.get("MM_TRF_DRIVER", "library")
        self._overlap_checked = False
        self._avoid_fused = False

    # -- reductions that need the cross-rank sum when sharded --
    def _ar(self, *tensors):
        if self.allreduce is not None:
            for t in tensors:
                self.allreduce(t)

    def _cost_dev(self, x):
        c2, _ = self.pb.residual(self._cams(x), self._pts(x))
        self._ar(c2)
        return c2

    def _cost(self, cams, pts):
        c2, _ = self.pb.residual(cams, pts)
        self._ar(c2)
        return 0.5 * float(c2.item())

    def _cams(self, v):
        return v[:self.nc].view(self.pb.F, 6)

    def _pts(self, v):
        return v[self.nc:].view(self.pb.P, 3)

    def _dots(self, pairs):
        """[<a, b> for (a, b) in pairs] as a device vector; the camera part of the parameter vector is replicated on
        every rank, the point part is sharded."""
        out = self.pb.multi_dot(pairs, self.nc)          # [k, 3] = (camera part, point part, total), one launch
        return self._combine(out)

    def _combine(self, rows):
        """rows [k, 3] of a fused pass -> [k] totals (the point part is summed across ranks when sharded)."""
        if self.allreduce is None:
            return rows[:, 2]
        # The camera part is replicated, but its reduction tree depends on the LOCAL vector length (shards differ), so
        # the ranks' copies differ in the last bit.  Rank 0's copy is the one that enters the sum: every rank then
        # holds the same scalars bit for bit, takes the same accept / reject / terminate decisions and stays inside the
        # same sequence of collectives.
        t = (rows[:, 0] + rows[:, 1]) if getattr(self.allreduce, "rank", 0) == 0 else rows[:, 1].clone()
        t = t.contiguous()
        self.allreduce(t)
        return t

    def _dots_sharded(self, pairs):
        """Inner products of residual-space vectors (every rank holds its own observations)."""
        out = self.pb.multi_dot(pairs, 0)[:, 2].contiguous()
        self._ar(out)
        return out

    def _normal(self, x, g):
        """Block normal equations at x; the gradient goes straight into the flat buffer g."""
        if self._native:      # the sweeps write B / C into the persistent blocks and g_c / g_p into the two halves of g
            out = (self._B, self._cams(g), self._C, self._pts(g))
            B, gc, C, gp = self.pb.normal_eq(self._cams(x), self._pts(x), out=out)
            self._ar(B, gc)
            return B, C
        B, gc, C, gp = self.pb.normal_eq(self._cams(x), self._pts(x))
        self._ar(B, gc)          # camera blocks are sums over all observations; point blocks are local
        g[:self.nc] = gc.reshape(-1)
        g[self.nc:] = gp.reshape(-1)
        return B, C

    def _scale_inv(self, B, C, old=None):
        if self._native:      # one launch: sqrt of the block diagonals, zeros -> 1 / running maximum, in place
            return self.pb.scale_update(B, C, self._si, old is None)
        si = torch.cat([torch.diagonal(B, dim1=1, dim2=2).reshape(-1), C[:, self.diag_idx].reshape(-1)]).sqrt_()
        if old is None:
            si[si == 0] = 1.0
            return si
        return torch.maximum(si, old)

    def _damped_blocks(self, B, C, si, reg):
        """B + reg diag(scale_inv^2), C + reg diag(scale_inv^2) (packed 6)."""
        if self._native:
            return self.pb.damp(B, C, si, reg, self._Bd, self._Cd)
        nc, F, P = self.nc, self.pb.F, self.pb.P
        sic2_36 = torch.diag_embed((si[:nc] * si[:nc]).view(F, 6))
        sip2_6 = torch.zeros((P, 6), dtype=si.dtype, device=si.device)
        sip2_6[:, self.diag_idx] = (si[nc:] * si[nc:]).view(P, 3)
        return torch.addcmul(B, sic2_36, reg), torch.addcmul(C, sip2_6, reg)

    def _serial_fallback(self, x, Bd, Cd, gc, gp, half_bw, solve=True):
        warnings.warn("mm_ba_schur_solve: kernels are being serialised; building and solving the reduced system one "
                      "after the other from here on")
        self.pb.overlap = False
        if solve:
            return self.pb.schur_solve(self._cams(x), self._pts(x), Bd, Cd, gc, gp, half_bw)

    def solve(self, cams0, pts0, ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, local_points_norm=None):
        pb = self.pb
        dev = pb.device
        F, P = pb.F, pb.P
        self.nc = nc = 6 * F
        n = nc + 3 * P
        f64 = dict(dtype=torch.float64, device=dev)
        x = torch.cat([cams0.reshape(-1), pts0.reshape(-1)]).to(**f64).contiguous()
        g = torch.empty(n, **f64)
        self.diag_idx = torch.tensor([0, 3, 5], device=dev)
        # block glue as single launches on persistent buffers when the problem object provides them (ops.BADevice)
        self._native = hasattr(pb, "damp") and hasattr(pb, "scale_update")
        if self._native:
            self._B, self._C = torch.empty((F, 6, 6), **f64), torch.empty((P, 6), **f64)
            self._Bd, self._Cd = torch.empty((F, 6, 6), **f64), torch.empty((P, 6), **f64)
            self._si = torch.empty(n, **f64)

        if (self.driver == "library" and self.allreduce is None and hasattr(pb, "trf_solve")
                and not getattr(pb, "overlap", False)):
            return self._solve_library(x, ftol, xtol, gtol, max_nfev, verbose)
        c2 = self._cost_dev(x)
        cost = 0.5 * float(c2.item())
        if not np.isfinite(cost):
            raise ValueError("Residuals are not finite in the initial point.")
        # band of the reduced camera system: |camera i - camera j| <= span  ->  |row - col| <= 6 span + 5
        span = torch.tensor([float(pb.cam_span)], **f64)
        if self.allreduce is not None:
            self.allreduce(span, op="max")
        span_all = int(span.item())
        half_bw = 6 * span_all + 5
        # Sharded: what is exchanged for the reduced camera system is decided from GLOBAL quantities only, so that every
        # rank enters the same collective with the same size whatever its own shard looks like (a rank whose points
        # include one very long track, or no observation at all, builds S with the general kernel; the others with the
        # pair-list kernel).  Every rank's S is zero outside the lower band |i - j| <= half_bw or symmetric inside it,
        # so the packed band [n, half_bw + 1] carries everything the factorisation reads.
        band_exchange = (self.allreduce is not None and hasattr(pb, "band_view")
                         and span_all <= getattr(pb, "max_band_span", 192) and half_bw < nc)
        nfev, njev = 1, 1
        if (self.driver == "library" and self.allreduce is not None and hasattr(pb, "trf_solve_dist")
                and getattr(self.allreduce, "world_size", 1) <= 16):
            # sharded: the same library loop, with this process group's all-reduce as its callback (mm_ba_trf_dist)
            return self._solve_library(x, ftol, xtol, gtol, max_nfev, verbose, dist=(half_bw, band_exchange))
        if self._native and hasattr(pb, "trf_step2d") and hasattr(pb, "schur_solve") and hasattr(pb, "jvp_dots"):
            return self._solve_device(x, g, cost, half_bw, band_exchange, ftol, xtol, gtol, max_nfev, verbose)
        B, C = self._normal(x, g)
        si = self._scale_inv(B, C)
        xs = x * si
        Delta = float(torch.sqrt(self._dots([(xs, xs)])[0]).item())
        if Delta == 0:
            Delta = 1.0
        if max_nfev is None:
            max_nfev = n * 100
        alpha = 0.0
        termination = None
        iteration = 0
        step_norm = None
        actual = None
        if verbose == 2:
            _print_header()
        # wall time between the host synchronisation points of an iteration (the syncs drain the stream, so these are
        # real intervals): normal equations, damping, Schur, Cholesky, subspace .. sync A | trial steps .. accept
        seg = {"to_syncA": 0.0, "syncA_to_accept": 0.0}
        t_mark = time.perf_counter()
        g_norm = None
        while True:
            # g_h = d * g (d = 1 / scale_inv) and d * g_h in one pass that also yields |g_h|^2 and |g|_inf
            gh, ghs = torch.empty_like(g), torch.empty_like(g)
            r0 = pb.trf_fused(0, [g, si], [gh, ghs], split=nc)
            gh2_t = self._combine(r0[:1]).contiguous()
            if self.allreduce is not None:
                gmax_p = r0[1, 1:2].contiguous()
                self.allreduce(gmax_p, op="max")
                gmax = torch.maximum(r0[1, 0:1], gmax_p)
            else:
                gmax = r0[1, 2:3]
            u1 = pb.jvp(self._cams(x), self._pts(x), self._cams(ghs), self._pts(ghs)).reshape(-1)   # J (d g_h)
            d11_t = self._dots_sharded([(u1, u1)])
            if termination is not None or nfev == max_nfev:
                g_norm = float(gmax.item())
                if verbose == 2:
                    _print_iteration(iteration, nfev, cost, actual, step_norm, g_norm)
                break
            # Cauchy-derived regulariser (trf.py:473-477) computed on the device: the reduced system is built and
            # factored without the host having seen |g|, |g_h| or |J_h g_h| (they arrive with sync A).
            # Damped blocks: J^T J + reg * D^-2  (D^-2 = scale_inv^2).  No gauge is fixed (as in the reference), so
            # J^T J has a 7-dimensional null space and only the damping makes the reduced system definite; SciPy's
            # LSMR copes with a singular system, a Cholesky factorisation needs `reg` to stay above rounding:
            # a floor of 1e-9 (relative to the unit diagonal of the scaled system), x100 on a bad pivot, and the
            # raised floor is kept for the rest of the solve (a system that needed it once needs it again).
            damp = pb.trf_damping(gh2_t, d11_t, Delta, self.min_damping)
            reg_eff = damp[1:2]
            gc, gp = self._cams(g), self._pts(g)
            for attempt in range(6):
                Bd, Cd = self._damped_blocks(B, C, si, reg_eff)
                if self.allreduce is None and hasattr(pb, "schur_solve"):
                    # one GPU: the build of S and its factorisation overlap (mm_ba_schur_solve)
                    info, v, Cinv = pb.schur_solve(self._cams(x), self._pts(x), Bd, Cd, gc, gp, half_bw)
                    if getattr(pb, "overlap", False) and not self._overlap_checked:
                        # first overlapped solve: make sure the two streams really ran concurrently (a profiler that
                        # serialises kernels starves the consumer, which then gives up with info = -1)
                        self._overlap_checked = True
                        if int(info.item()) < 0:
                            info, v, Cinv = self._serial_fallback(x, Bd, Cd, gc, gp, half_bw)
                else:
                    S, v, Cinv = pb.schur(self._cams(x), self._pts(x), Bd, Cd, gc, gp)
                if self.allreduce is not None:
                    # every rank added the full blockdiag(Bd) and gc: remove the duplicates after the sum
                    ws = self.allreduce.world_size
                    if band_exchange:
                        # only the lower band is populated (pair-list Schur kernel): exchange n x (hb + 1) doubles
                        # (12.7 MB at 500 cameras) instead of the dense 72 MB
                        band = pb.band_view(half_bw)
                        packed = band.contiguous()
                        self.allreduce(packed)
                        band.copy_(packed)
                    else:
                        self.allreduce(S)
                    self.allreduce(v)
                    if ws > 1:
                        blk = S.reshape(F, 6, F, 6)
                        f = torch.arange(F, device=dev)
                        blk[f, :, f, :] -= (ws - 1) * Bd
                        v -= (ws - 1) * gc.reshape(-1)
                if self.allreduce is not None or not hasattr(pb, "schur_solve"):
                    if hasattr(pb, "chol_solve_sym"):
                        # after a band exchange only the lower band is the sum; the dense exchange sums everything
                        info = pb.chol_solve_sym(S, v, half_bw, both_triangles=not band_exchange)
                    else:
                        info = pb.chol_solve(S, v, half_bandwidth=half_bw)
                # q = [v ; dp] = (J^T J + reg D^-2)^-1 g, the unscaled Gauss-Newton step
                dp = pb.backsub(self._cams(x), self._pts(x), Cinv, gp, v.view(F, 6)).reshape(-1)
                # orthonormal basis of span{g_h, gn_h} (trf.py:481-482) in three fused passes:
                #   gn_h = q * scale_inv, q1 = g_h / |g_h|            (+ <q1, gn_h>, |gn_h|^2)
                #   w = gn_h - <q1, gn_h> q1                            (+ |w|^2)
                #   q2 = w / |w|, s1 = d q1, s2 = d q2 (unscaled steps) (+ the five step inner products)
                gn, q1, w = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
                r1 = self._combine(pb.trf_fused(1, [v, dp, si, gh], [gn, q1], [gh2_t], split=nc)[:2])
                sc, gn2 = r1[0:1].contiguous(), r1[1]
                wn2 = self._combine(pb.trf_fused(2, [gn, q1], [w], [sc], split=nc)[:1]).contiguous()
                q2, s1, s2 = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
                nn = self._combine(pb.trf_fused(3, [w, q1, si, gh, x], [q2, s1, s2], [wn2], split=nc)[:5])
                wn2 = wn2[0]
                Jq2 = pb.jvp(self._cams(x), self._pts(x), self._cams(s2), self._pts(s2)).reshape(-1)
                # J_h q1 = J (d q1) = u1 / |g_h|: <Jq1, Jq1> = d11 / |g_h|^2, <Jq1, Jq2> = <u1, Jq2> / |g_h|
                bs = self._dots_sharded([(u1, Jq2), (Jq2, Jq2)])
                # ---- host sync A ----
                vals = torch.cat([info.to(torch.float64), wn2.reshape(1), gn2.reshape(1), bs, nn, gmax,
                                  gh2_t.reshape(1), d11_t.reshape(1), reg_eff]).tolist()
                if int(vals[0]) == 0:
                    break
                if int(vals[0]) < 0:
                    # the single-launch factorisation gave up waiting (its workgroups or the producer of S were not
                    # co-resident: another tenant, a profiler attaching mid-run).  One GPU: redo this attempt with the
                    # build and the solve one after the other and stay there; otherwise there is nothing to fall back to.
                    if self.allreduce is None and getattr(pb, "overlap", False):
                        self._serial_fallback(x, Bd, Cd, gc, gp, half_bw, solve=False)
                        continue
                    raise MMError("mm_chol_solve: the fused banded factorisation was abandoned (info = -1)")
                reg_eff = reg_eff * 100.0
                if vals[-1] <= self.min_damping * (1.0 + 1e-12):      # failed AT the floor: the floor was too low
                    self.min_damping *= 100.0
            else:
                raise MMError(f"reduced camera system is not positive definite (pivot {int(vals[0])})")
            _, wn2, gn2, u1Jq2, b22, n11, n12, n22, g2, xx, g_norm, gh2, d11, _ = vals
            gh_norm = np.sqrt(gh2)
            b11, b12 = d11 / gh2, u1Jq2 / gh_norm
            t_now = time.perf_counter()
            seg["to_syncA"] += t_now - t_mark
            t_mark = t_now
            if g_norm < gtol:                              # (checked before the step is used, as trf.py:443 does)
                termination = 1
            if verbose == 2:
                _print_iteration(iteration, nfev, cost, actual, step_norm, g_norm)
            if termination is not None:
                break
            if not (wn2 > 1e-28 * max(gn2, 1e-300)):       # gn_h parallel to g_h: the subspace is one-dimensional
                s2 = torch.zeros_like(s1)
                b12, b22, n12, n22, g2 = 0.0, 1.0, 0.0, 0.0, 0.0
            B_S = np.array([[b11, b12], [b12, b22]])
            g_S = np.array([gh_norm, g2])
            x_norm = np.sqrt(xx)
            actual = -1.0
            while actual <= 0 and nfev < max_nfev:
                p_S, _ = _solve_trust_region_2d(B_S, g_S, Delta)
                predicted = -(0.5 * p_S @ B_S @ p_S + g_S @ p_S)
                x_new = torch.empty_like(x)
                pb.trf_fused(4, [x, s1, s2], [x_new], h0=float(p_S[0]), h1=float(p_S[1]), split=nc)
                cost_new = 0.5 * float(self._cost_dev(x_new).item())      # ---- host sync 3 (per trial step) ----
                nfev += 1
                step_h_norm = float(_norm(p_S))
                if not np.isfinite(cost_new):
                    Delta = 0.25 * step_h_norm
                    continue
                actual = cost - cost_new
                Delta_new, ratio = _update_tr_radius(Delta, actual, predicted, step_h_norm, step_h_norm > 0.95 * Delta)
                step_norm = float(np.sqrt(max(p_S[0] ** 2 * n11 + 2 * p_S[0] * p_S[1] * n12 + p_S[1] ** 2 * n22, 0.0)))
                termination = _check_termination(actual, cost, step_norm, x_norm, ratio, ftol, xtol)
                if termination is not None:
                    break
                alpha *= Delta / Delta_new
                Delta = Delta_new
            t_now = time.perf_counter()
            seg["syncA_to_accept"] += t_now - t_mark
            t_mark = t_now
            if actual > 0:
                x, cost = x_new, cost_new
                B, C = self._normal(x, g)
                njev += 1
                si = self._scale_inv(B, C, si)
            else:
                step_norm = 0
                actual = 0
            iteration += 1
        if termination is None:
            termination = 0
        return BAResult(cams=self._cams(x).clone(), pts=self._pts(x).clone(), cost=cost, optimality=g_norm, nfev=nfev,
                        njev=njev, status=termination, message=_MESSAGES[termination], success=termination > 0,
                        iterations=iteration, host_segments_ms={k: 1e3 * v for k, v in seg.items()})


def _solve_library(self, x, ftol, xtol, gtol, max_nfev, verbose, dist=None):
    """The whole loop of `_solve_device` inside the library: mm_ba_trf on one GPU, mm_ba_trf_dist (dist = (half bandwidth,
    band exchange), both decided from all-reduced quantities) when the points are sharded over ranks (csrc/trf.hip)."""
    nc = self.nc
    cams, pts = x[:nc], x[nc:]
    t0 = time.perf_counter()
    if dist is None:
        rep, rows = self.pb.trf_solve(cams, pts, ftol, xtol, gtol, max_nfev, self.min_damping,
                                      log_cap=4096 if verbose == 2 else 0)
    else:
        rep, rows = self.pb.trf_solve_dist(cams, pts, ftol, xtol, gtol, self.allreduce, dist[0], dist[1], max_nfev,
                                           self.min_damping, log_cap=4096 if verbose == 2 else 0)
    self.min_damping = rep.min_damping
    if verbose == 2:
        _print_header()
        for it, nf, c, red, stp, opt in rows:
            _print_iteration(it, nf, c, None if np.isnan(red) else red, None if np.isnan(stp) else stp, opt)
        if rep.log_rows > len(rows):
            print(f"... ({rep.log_rows - len(rows)} more iterations)")
    return BAResult(cams=self._cams(x).clone(), pts=self._pts(x).clone(), cost=rep.cost, optimality=rep.optimality,
                    nfev=rep.nfev, njev=rep.njev, status=rep.status, message=_MESSAGES[rep.status],
                    success=rep.status > 0, iterations=rep.iterations, chol_fallbacks=rep.chol_fallbacks,
                    collectives=rep.collectives,
                    host_segments_ms={"library": 1e3 * (time.perf_counter() - t0)})


SchurTRF._solve_library = _solve_library


def _solve_device(self, x, g, cost, half_bw, band_exchange, ftol, xtol, gtol, max_nfev, verbose):
    """The same iteration as `SchurTRF.solve`'s generic loop (which serves the CPU stand-in of the tests), with everything
    device resident: the 2-D trust-region subproblem is solved by a kernel from the fused passes' results
    (mm_trf_step2d), the trial point is formed from its output and the host reads one small board of scalars per trial
    step -- after the trial cost is known -- instead of synchronising twice per iteration.  Sharded (points partitioned
    over ranks): every sum over observations / points is all-reduced on the device before the next kernel reads it (the
    "total" column of the fused passes' result rows is overwritten with the cross-rank total), so all ranks feed
    identical scalars to identical kernels and read identical boards."""
    pb = self.pb
    ar = self.allreduce

    def fix_params(rows, k, max_row=None):
        """result rows of a pass over the PARAMETER vector: column 2 <- cross-rank total (camera part replicated)."""
        if ar is None:
            return rows
        rows[:k, 2] = self._combine(rows[:k])
        if max_row is not None:
            m = rows[max_row, 1:2].contiguous()
            ar(m, op="max")
            rows[max_row, 2:3] = torch.maximum(rows[max_row, 0:1], m)
        return rows

    def fix_residual(rows):
        """result rows of inner products of RESIDUAL-space vectors (every rank holds its own observations)."""
        if ar is None:
            return rows
        t = rows[:, 2].contiguous()
        ar(t)
        rows[:, 2] = t
        return rows

    def reduced_solve(x, Bd, Cd, gc, gp):
        """-> (info, v = solution of the reduced camera system, Cinv) at the current iterate x."""
        if ar is None:
            return pb.schur_solve(self._cams(x), self._pts(x), Bd, Cd, gc, gp, half_bw)
        S, v, Cinv = pb.schur(self._cams(x), self._pts(x), Bd, Cd, gc, gp)
        ws = ar.world_size
        if band_exchange:
            # only the lower band is exchanged: n x (hb + 1) doubles (12.7 MB at 500 cameras) instead of the dense 72 MB
            band = pb.band_view(half_bw)
            packed = band.contiguous()
            ar(packed)
            band.copy_(packed)
        else:
            ar(S)
        ar(v)
        if ws > 1:      # every rank added the full blockdiag(Bd) and gc: remove the duplicates after the sum
            blk = S.reshape(pb.F, 6, pb.F, 6)
            f = torch.arange(pb.F, device=pb.device)
            blk[f, :, f, :] -= (ws - 1) * Bd
            v -= (ws - 1) * gc.reshape(-1)
        info = pb.chol_solve_sym(S, v, half_bw, both_triangles=not band_exchange)
        return info, v, Cinv

    # an abandoned single-launch factorisation switches THIS solve to the launch-per-column path; the context's setting
    # (normally the shared default context) and this object's flag are restored on the way out, as trf.hip's FusedGuard does
    self._avoid_fused = False
    prev_avoid = pb.ctx.control(pb.ctx.CTL_CHOL_AVOID_FUSED, -1) if hasattr(pb, "ctx") else None
    try:
        return _solve_device_loop(self, x, g, cost, half_bw, ftol, xtol, gtol, max_nfev, verbose, fix_params, fix_residual,
                                  reduced_solve)
    finally:
        if prev_avoid is not None:
            pb.ctx.control(pb.ctx.CTL_CHOL_AVOID_FUSED, prev_avoid)
        self._avoid_fused = False


def _solve_device_loop(self, x, g, cost, half_bw, ftol, xtol, gtol, max_nfev, verbose, fix_params, fix_residual, reduced_solve):
    pb = self.pb
    ar = self.allreduce
    F, P, nc = pb.F, pb.P, self.nc
    n = nc + 3 * P
    f64 = dict(dtype=torch.float64, device=pb.device)
    cams, pts = self._cams, self._pts
    nfev, njev = 1, 1
    B, C = self._normal(x, g)
    si = self._scale_inv(B, C)
    xs = x * si
    Delta = float(torch.sqrt(fix_params(pb.multi_dot([(xs, xs)], nc), 1)[0, 2]).item())
    del xs
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = n * 100
    gh, ghs, gn, q1, w, q2, s1, s2, x_new = (torch.empty_like(g) for _ in range(9))
    board = torch.zeros(16, **f64)
    cost_slot = board[14:15]
    alpha = 0.0
    termination, iteration, step_norm, actual, g_norm = None, 0, None, None, None
    if verbose == 2:
        _print_header()
    seg = {"to_syncA": 0.0, "syncA_to_accept": 0.0}
    t_mark = time.perf_counter()
    while True:
        r0 = fix_params(pb.trf_fused(0, [g, si], [gh, ghs], split=nc), 1, max_row=1)   # rows: |g_h|^2 ; max |g|
        gh2_t = r0[0, 2:3]
        u1, d11 = pb.jvp_dots(cams(x), pts(x), cams(ghs), pts(ghs))         # J (d g_h) and |J d g_h|^2 in one sweep
        d11 = fix_residual(d11)
        if termination is not None or nfev == max_nfev:
            g_norm = float(r0[1, 2].item())
            if verbose == 2:
                _print_iteration(iteration, nfev, cost, actual, step_norm, g_norm)
            break
        damp = pb.trf_damping(gh2_t, d11[0, 2:3], Delta, self.min_damping)
        reg_eff = damp[1:2]
        gc, gp = cams(g), pts(g)
        vals = None
        for attempt in range(6):
            Bd, Cd = self._damped_blocks(B, C, si, reg_eff)
            info, v, Cinv = reduced_solve(x, Bd, Cd, gc, gp)
            dp = pb.backsub(cams(x), pts(x), Cinv, gp, v.view(F, 6)).reshape(-1)
            # orthonormal basis of span{g_h, gn_h} (trf.py:481-482) in three fused passes (see the generic loop)
            r1 = fix_params(pb.trf_fused(1, [v, dp, si, gh], [gn, q1], [gh2_t], split=nc), 2)
            r2 = fix_params(pb.trf_fused(2, [gn, q1], [w], [r1[0, 2:3]], split=nc), 1)
            r3 = fix_params(pb.trf_fused(3, [w, q1, si, gh, x], [q2, s1, s2], [r2[0, 2:3]], split=nc), 5)
            _, bs = pb.jvp_dots(cams(x), pts(x), cams(s2), pts(s2), other=u1)   # <J s2, u1>, |J s2|^2
            bs = fix_residual(bs)

            def trial(Delta_):
                pb.trf_step2d(r0, d11, r1, r2, r3, bs, reg_eff, info, Delta_, board)
                pb.trf_fused(5, [x, s1, s2], [x_new], [board], split=nc)
                pb.residual(cams(x_new), pts(x_new), cost_out=cost_slot)
                if ar is not None:
                    ar(cost_slot)
                return board.tolist()                                     # ---- the host sync of a trial step ----

            vals = trial(Delta)          # enqueued before the host knows whether the factorisation succeeded
            inf = int(vals[6])
            if inf == 0:
                break
            if inf < 0:
                # the single-launch factorisation gave up waiting (see the generic loop): build and solve one after
                # the other from here on
                if ar is None and getattr(pb, "overlap", False):
                    self._serial_fallback(x, Bd, Cd, gc, gp, half_bw, solve=False)
                    continue
                if not self._avoid_fused and hasattr(pb, "ctx"):
                    # its workgroups were not co-resident (another tenant on the GPU): repeat the attempt, same
                    # damping, on the launch-per-column factorisation and stay there for the rest of the solve.
                    # (Sharded: every rank reads the same replicated board... but info is LOCAL -- a rank-local decision
                    # would desynchronise the collectives, so the switch is only taken on one GPU.)
                    if ar is None:
                        self._avoid_fused = True
                        pb.ctx.control(pb.ctx.CTL_CHOL_AVOID_FUSED, 1)
                        continue
                raise MMError("mm_chol_solve: the fused banded factorisation was abandoned (info = -1)")
            if vals[13] <= self.min_damping * (1.0 + 1e-12):      # failed AT the floor: the floor was too low
                self.min_damping *= 100.0
            reg_eff = reg_eff * 100.0
        else:
            raise MMError(f"reduced camera system is not positive definite (pivot {int(vals[6])})")
        g_norm, xx = vals[10], vals[9]
        t_now = time.perf_counter()
        seg["to_syncA"] += t_now - t_mark
        t_mark = t_now
        if g_norm < gtol:                              # (checked before the step is used, as trf.py:443 does; the
            termination = 1                            # trial point enqueued above is simply dropped)
        if verbose == 2:
            _print_iteration(iteration, nfev, cost, actual, step_norm, g_norm)
        if termination is not None:
            break
        x_norm = np.sqrt(xx)
        actual = -1.0
        have = True
        while actual <= 0 and nfev < max_nfev:
            if not have:
                vals = trial(Delta)
            have = False
            predicted, step_h_norm, step_norm_dev = vals[2], vals[3], vals[4]
            cost_new = 0.5 * vals[14]
            nfev += 1
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual = cost - cost_new
            Delta_new, ratio = _update_tr_radius(Delta, actual, predicted, step_h_norm, step_h_norm > 0.95 * Delta)
            step_norm = float(step_norm_dev)
            termination = _check_termination(actual, cost, step_norm, x_norm, ratio, ftol, xtol)
            if termination is not None:
                break
            alpha *= Delta / Delta_new
            Delta = Delta_new
        t_now = time.perf_counter()
        seg["syncA_to_accept"] += t_now - t_mark
        t_mark = t_now
        if actual > 0:
            x, x_new = x_new, x
            cost = cost_new
            B, C = self._normal(x, g)
            njev += 1
            si = self._scale_inv(B, C, si)
        else:
            step_norm = 0
            actual = 0
        iteration += 1
    if termination is None:
        termination = 0
    return BAResult(cams=cams(x).clone(), pts=pts(x).clone(), cost=cost, optimality=g_norm, nfev=nfev,
                    njev=njev, status=termination, message=_MESSAGES[termination], success=termination > 0,
                    iterations=iteration, host_segments_ms={k: 1e3 * v for k, v in seg.items()})


SchurTRF._solve_device = _solve_device


def _finish_verbose(res, cost0, verbose):
    if verbose >= 1:
        print(res.message)
        print(f"Function evaluations {res.nfev}, initial cost {cost0:.4e}, final cost {res.cost:.4e}, "
              f"first-order optimality {res.optimality:.2e}.")


def reformatPointResult(result, n_frames, n_points):
    """x -> (points [P,3], list of F 4x4 extrinsics) (bundleAdjuster.py:137-157)."""
    x = np.asarray(result.x, float)
    points = x[n_frames * 6:].reshape((n_points, 3))
    frames = x[:n_frames * 6].reshape((n_frames, 6))
    extrinsics = []
    for rvec, tvec in zip(frames[:, :3], frames[:, 3:]):
        E = np.eye(4)
        E[:3, :3] = _rodrigues_matrix(rvec)
        E[:3, 3] = tvec
        extrinsics.append(E)
    return points, extrinsics


def reformatPoseResult(result, n_frames):
    """x -> list of F 3x4 extrinsics (bundleAdjuster.py:197-203)."""
    fp = np.asarray(result.x, float).reshape((n_frames, 6))
    return [np.hstack((_rodrigues_matrix(r), t.reshape(3, 1))) for r, t in zip(fp[:, :3], fp[:, 3:6])]


def solvePoints(frame_extrinsic_matrices, camera_intrinsic_matrix, points_3D, points_2D, frame_indices, point_indices,
                ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=2):
    """adjustPoints with the optimiser settings exposed; returns the full result object (x, cost, nfev, ...)."""
    ctx = default_context()
    dev = ctx.device
    ext = np.asarray(frame_extrinsic_matrices, float)
    F = len(ext)
    pts0 = np.asarray(points_3D, float).reshape(-1, 3)
    P = len(pts0)
    with np.errstate(all="ignore"):
        cams0 = frameParameters(ext).reshape(F, 6)
    pb = ops.BADevice(camera_intrinsic_matrix, frame_indices, point_indices, points_2D, F, P, dev, ctx)
    solver = SchurTRF(pb)
    cams_d, pts_d = _dev(cams0, dev), _dev(pts0, dev)
    cost0 = None
    if verbose >= 1:
        cost0 = solver._cost(cams_d, pts_d)
    res = solver.solve(cams_d, pts_d, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, verbose=verbose)
    res.x = np.concatenate([res.cams.cpu().numpy().reshape(-1), res.pts.cpu().numpy().reshape(-1)])
    if verbose >= 1:
        _finish_verbose(res, cost0, verbose)
    return res


def adjustPoints(frame_extrinsic_matrices, camera_intrinsic_matrix, points_3D, points_2D, frame_indices, point_indices):
    """Full bundle adjustment over all cameras and points (bundleAdjuster.py:160-194) with the reference's settings
    (x_scale='jac', ftol=1e-4, verbose=2 progress table).  -> (points [P,3], list of F 4x4 extrinsics)."""
    res = solvePoints(frame_extrinsic_matrices, camera_intrinsic_matrix, points_3D, points_2D, frame_indices,
                      point_indices, ftol=1e-4, verbose=2)
    F = len(frame_extrinsic_matrices)
    return reformatPointResult(res, F, len(np.asarray(points_3D).reshape(-1, 3)))


# ----------------------------------------------------------------------------------------------- pose-only refinement

def _solve_lsq_trust_region_eig(lam, vg, V, Delta, m, initial_alpha, rtol=0.01, max_iter=10):
    """scipy/optimize/_lsq/common.py:solve_lsq_trust_region expressed through the eigen-decomposition of J^T J.
    The pose-only Jacobian is block diagonal (one 2n x 6 block per camera), so its SVD is the union of the per-camera
    SVDs: singular values s = sqrt(eig(B_f)), right vectors V_f, and s*U^T f = V^T g.
    lam [F,6] eigenvalues, vg [F,6] = V^T g, V [F,6,6]."""
    s2 = np.maximum(lam, 0.0).ravel()
    s = np.sqrt(s2)
    suf = vg.ravel()
    n = s.size

    def phi_and_derivative(alpha):
        denom = s2 + alpha
        p_norm = _norm(suf / denom)
        return p_norm - Delta, -np.sum(suf ** 2 / denom ** 3) / p_norm

    def back(coef):
        return -np.einsum("fij,fj->fi", V, coef.reshape(V.shape[0], 6))

    full_rank = (m >= n) and (s.min() > EPS * m * s.max())
    if full_rank:
        p = back(suf / s2)
        if _norm(p) <= Delta:
            return p, 0.0, 0
    alpha_upper = _norm(suf) / Delta
    if full_rank:
        phi, phi_prime = phi_and_derivative(0.0)
        alpha_lower = -phi / phi_prime
    else:
        alpha_lower = 0.0
    if initial_alpha is None or (not full_rank and initial_alpha == 0):
        alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
    else:
        alpha = initial_alpha
    it = 0
    for it in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        phi, phi_prime = phi_and_derivative(alpha)
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if np.abs(phi) < rtol * Delta:
            break
    p = back(suf / (s2 + alpha))
    p *= Delta / _norm(p)
    return p, alpha, it + 1


def solvePose(frame_extrinsic_matrices, camera_intrinsic_matrix, points_2D, ftol=1e-4, xtol=1e-8, gtol=1e-8,
              max_nfev=None, verbose=2):
    """adjustPose with the optimiser exposed.  The reference calls least_squares(poseFun, ..., ftol=1e-4) with every
    other setting at its default (bundleAdjuster.py:232-241): TRF, dense 2-point Jacobian, tr_solver='exact' (SVD),
    x_scale=1.  Same iteration here; the camera blocks B_f = J_f^T J_f and g_f come from the HIP sweep with the
    points held fixed, and the 'exact' trust-region solve works on their 6x6 eigen-decompositions."""
    ctx = default_context()
    dev = ctx.device
    ext = np.asarray(frame_extrinsic_matrices, float)
    F = len(ext)
    pattern_size = int(len(points_2D) / F)
    pts3 = np.zeros((pattern_size, 3), np.float32)           # the (4,3) chessboard of side 2, bundleAdjuster.py:220-223
    grid = np.mgrid[0:4, 0:3].T.reshape(-1, 2) * 2
    pts3[:, 0] = grid[:, 0]
    pts3[:, 2] = grid[:, 1]
    fi = np.repeat(np.arange(F), pattern_size)
    pi = np.repeat([np.arange(pattern_size)], F, axis=0).reshape(pattern_size * F)
    with np.errstate(all="ignore"):
        x = frameParameters(ext).reshape(F, 6)
    pb = ops.BADevice(camera_intrinsic_matrix, fi, pi, points_2D, F, pattern_size, dev, ctx, pairs=False)
    pts_d = _dev(pts3.astype(np.float64), dev)
    m, n = 2 * len(fi), 6 * F

    def fun(xc):
        c2, _ = pb.residual(_dev(xc, dev), pts_d)
        return 0.5 * float(c2.item())

    def blocks(xc):
        B, g, _, _ = pb.normal_eq(_dev(xc, dev), pts_d, want_cams=True, want_pts=False)
        return B.cpu().numpy(), g.cpu().numpy()

    cost = fun(x)
    cost0 = cost
    if not np.isfinite(cost):
        raise ValueError("Residuals are not finite in the initial point.")
    nfev, njev = 1, 1
    B, g = blocks(x)
    Delta = _norm(x)          # x_scale = 1  (trf.py:427-430)
    if Delta == 0:
        Delta = 1.0
    if max_nfev is None:
        max_nfev = n * 100
    alpha = 0.0
    termination, iteration, step_norm, actual = None, 0, None, None
    if verbose == 2:
        _print_header()
    while True:
        g_norm = float(np.abs(g).max())
        if g_norm < gtol:
            termination = 1
        if verbose == 2:
            _print_iteration(iteration, nfev, cost, actual, step_norm, g_norm)
        if termination is not None or nfev == max_nfev:
            break
        lam, V = np.linalg.eigh(B)
        vg = np.einsum("fji,fj->fi", V, g)
        actual = -1.0
        while actual <= 0 and nfev < max_nfev:
            step, alpha, _ = _solve_lsq_trust_region_eig(lam, vg, V, Delta, m, alpha)
            predicted = -(0.5 * np.einsum("fi,fij,fj->", step, B, step) + np.sum(g * step))
            x_new = x + step
            cost_new = fun(x_new)
            nfev += 1
            step_h_norm = _norm(step)
            if not np.isfinite(cost_new):
                Delta = 0.25 * step_h_norm
                continue
            actual = cost - cost_new
            Delta_new, ratio = _update_tr_radius(Delta, actual, predicted, step_h_norm, step_h_norm > 0.95 * Delta)
            step_norm = step_h_norm
            termination = _check_termination(actual, cost, step_norm, _norm(x), ratio, ftol, xtol)
            if termination is not None:
                break
            alpha *= Delta / Delta_new
            Delta = Delta_new
        if actual > 0:
            x, cost = x_new, cost_new
            B, g = blocks(x)
            njev += 1
        else:
            step_norm = 0
            actual = 0
        iteration += 1
    if termination is None:
        termination = 0
    res = BAResult(x=x.reshape(-1), cost=cost, optimality=g_norm, nfev=nfev, njev=njev, status=termination,
                   message=_MESSAGES[termination], success=termination > 0)
    if verbose >= 1:
        _finish_verbose(res, cost0, verbose)
    return res


def adjustPose(frame_extrinsic_matrices, camera_intrinsic_matrix, points_2D):
    """Pose-only refinement against the fixed chessboard (bundleAdjuster.py:214-243) -> list of F 3x4 extrinsics."""
    res = solvePose(frame_extrinsic_matrices, camera_intrinsic_matrix, points_2D, ftol=1e-4, verbose=2)
    return reformatPoseResult(res, len(frame_extrinsic_matrices))
