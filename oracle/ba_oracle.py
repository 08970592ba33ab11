"""CPU ORACLE (test infrastructure only) — NumPy/SciPy restatement of the reference's
bundle-adjustment, triangulation and track-bookkeeping path.

    *** This module is the CHECKER.  Only tests/, __graft_entry__.smoke() and bench.py's
    *** cpu_baseline leg may import it.  Nothing under meatmodeler_amd/ may.

Pinned against golden vectors G1-G8 (tests/golden/, captured by importing the reference's own
NumPy/SciPy code in the build container: tests/golden/make_golden.py) by tests/test_oracle_golden.py.
`triangulate_dlt` restates OpenCV's cv2.triangulatePoints (opencv-python~=4.5.2.54, absent offline,
no reference fixture) from its published algorithm: **parity unpinned** for that function.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.sparse import csr_matrix


# ------------------------------------------------------------------ cost model (bundleAdjuster.py)

def rotate(points, rvecs):
    """Rodrigues rotation of points[n,3] by axis-angle rvecs[n,3] (bundleAdjuster.py:7-28).

    theta = |r|; unit axis k = r/theta with 0/0 -> 0 (the reference's nan_to_num), so theta == 0
    leaves the point unchanged.  X' = cos X + sin (k x X) + (1-cos)(k.X) k.
    """
    points = np.asarray(points, float)
    rvecs = np.asarray(rvecs, float)
    theta = np.sqrt((rvecs * rvecs).sum(axis=1, keepdims=True))
    safe = np.where(theta == 0.0, 1.0, theta)
    k = np.where(theta == 0.0, 0.0, rvecs / safe)
    c, s = np.cos(theta), np.sin(theta)
    kdotx = (k * points).sum(axis=1, keepdims=True)
    kxx = np.stack([k[:, 1] * points[:, 2] - k[:, 2] * points[:, 1],
                    k[:, 2] * points[:, 0] - k[:, 0] * points[:, 2],
                    k[:, 0] * points[:, 1] - k[:, 1] * points[:, 0]], axis=1)
    return c * points + s * kxx + kdotx * (1.0 - c) * k


def project(points, frame_params, K):
    """points[n,3], frame_params[n,6]=(r,t), full 3x3 K -> pixels[n,2] (bundleAdjuster.py:31-52)."""
    Xc = rotate(points, frame_params[:, :3]) + frame_params[:, 3:6]
    u = Xc @ np.asarray(K, float).T            # einsum("ij,...j") of the reference, :47
    return u[:, :2] / u[:, 2:3]


def point_fun(x, K, n_frames, n_points, fi, pi, obs):
    """Residual vector [2*O], interleaved x,y (bundleAdjuster.py:81-102).
    x = [cam0(r,t) .. cam_{F-1}, pt0 .. pt_{P-1}] (:96-97)."""
    cams = x[:6 * n_frames].reshape(n_frames, 6)
    pts = x[6 * n_frames:].reshape(n_points, 3)
    return (project(pts[pi], cams[fi], K) - obs).ravel()


def pose_fun(x, K, n_frames, fi, pi, pts3, obs):
    """Pose-only residuals, points fixed (bundleAdjuster.py:206-211)."""
    cams = x.reshape(n_frames, 6)
    return (project(pts3[pi], cams[fi], K) - obs).ravel()


def frame_parameters(ext):
    """[F,3|4,4] extrinsics -> [6F] (r,t) rows (bundleAdjuster.py:105-134).

    theta = arccos((tr R - 1)/2) with NO clipping (:117-119); axis from the skew part over
    2 sin(theta) (:124-126); 0/0 -> 0 via nan_to_num (:131); r = axis * theta.
    """
    ext = np.asarray(ext, float)
    R = ext[:, :3, :3]
    with np.errstate(invalid="ignore", divide="ignore"):
        theta = np.arccos((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0) / 2.0)
        den = 2.0 * np.sin(theta)
        axis = np.stack([(R[:, 2, 1] - R[:, 1, 2]) / den,
                         (R[:, 0, 2] - R[:, 2, 0]) / den,
                         (R[:, 1, 0] - R[:, 0, 1]) / den], axis=1)
        rv = np.nan_to_num(axis) * theta[:, None]
    return np.hstack([rv, ext[:, :3, 3]]).reshape(-1)


def sparsity_pattern(n_frames, n_points, fi, pi):
    """CSR pattern of the Jacobian: rows 2i,2i+1 carry the 6 camera + 3 point columns
    (bundleAdjuster.py:55-78)."""
    fi = np.asarray(fi)
    pi = np.asarray(pi)
    O = fi.size
    cols = np.concatenate([6 * fi[:, None] + np.arange(6)[None, :],
                           6 * n_frames + 3 * pi[:, None] + np.arange(3)[None, :]], axis=1)  # [O,9]
    cols = np.repeat(cols, 2, axis=0)                                                          # [2O,9]
    indptr = np.arange(0, 18 * O + 1, 9)
    return csr_matrix((np.ones(cols.size, dtype=int), cols.ravel(), indptr),
                      shape=(2 * O, 6 * n_frames + 3 * n_points))


def rodrigues_matrix(rvec):
    """Closed-form axis-angle -> R (what cv2.Rodrigues computes at bundleAdjuster.py:153,201;
    OpenCV itself is absent: mathematical definition)."""
    r = np.asarray(rvec, float).reshape(3)
    th = float(np.sqrt(r @ r))
    if th == 0.0:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(th) * Kx + (1.0 - np.cos(th)) * (Kx @ Kx)


def reformat_point_result(x, n_frames, n_points):
    """x -> (points[P,3], list of F 4x4) (bundleAdjuster.py:137-157)."""
    cams = x[:6 * n_frames].reshape(n_frames, 6)
    pts = x[6 * n_frames:].reshape(n_points, 3)
    ext = []
    for f in range(n_frames):
        E = np.eye(4)
        E[:3, :3] = rodrigues_matrix(cams[f, :3])
        E[:3, 3] = cams[f, 3:]
        ext.append(E)
    return pts, ext


def adjust_points(ext, K, points_3D, points_2D, fi, pi, ftol=1e-4, xtol=1e-8, gtol=1e-8, verbose=0,
                  max_nfev=None, return_result=False):
    """Full BA exactly as the reference drives SciPy (bundleAdjuster.py:160-194):
    TRF, jac_sparsity, x_scale='jac', ftol=1e-4, 2-point finite differences, LSMR."""
    F, P = len(ext), len(points_3D)
    fi = np.asarray(fi)
    pi = np.asarray(pi)
    x0 = np.hstack([frame_parameters(ext), np.asarray(points_3D, float).reshape(3 * P)])
    A = sparsity_pattern(F, P, fi, pi)
    res = least_squares(point_fun, x0, jac_sparsity=A, verbose=verbose, x_scale="jac", ftol=ftol, xtol=xtol,
                        gtol=gtol, method="trf", max_nfev=max_nfev,
                        args=(np.asarray(K, float), F, P, fi, pi, np.asarray(points_2D, float)))
    out = reformat_point_result(res.x, F, P)
    return (out + (res,)) if return_result else out


def chessboard_points(pattern_size=12):
    """The (4,3) chessboard of side 2 in the x-z plane (bundleAdjuster.py:220-223)."""
    pts = np.zeros((pattern_size, 3))
    g = np.mgrid[0:4, 0:3].T.reshape(-1, 2) * 2
    pts[:, 0] = g[:, 0]
    pts[:, 2] = g[:, 1]
    return pts


def adjust_pose(ext, K, points_2D, ftol=1e-4, verbose=0, return_result=False):
    """Pose-only refinement (bundleAdjuster.py:214-243): dense TRF (exact SVD solver), ftol=1e-4."""
    F = len(ext)
    n = int(len(points_2D) / F)
    pts3 = chessboard_points(n)
    fi = np.repeat(np.arange(F), n)
    pi = np.tile(np.arange(n), F)
    x0 = frame_parameters(ext)
    res = least_squares(pose_fun, x0, verbose=verbose, ftol=ftol,
                        args=(np.asarray(K, float), F, fi, pi, pts3, np.asarray(points_2D, float)))
    cams = res.x.reshape(F, 6)
    out = [np.hstack([rodrigues_matrix(c[:3]), c[3:6].reshape(3, 1)]) for c in cams]
    return (out, res) if return_result else out


def jacobian_fd(x, K, n_frames, n_points, fi, pi, obs, h=1e-6):
    """Central-difference blocks (Jc[O,2,6], Jp[O,2,3]) — the checker for the analytic Jacobian the
    HIP path uses in place of SciPy's 2-point scheme (scipy/optimize/_numdiff.py:628-705)."""
    O = len(fi)
    cams = x[:6 * n_frames].reshape(n_frames, 6)
    pts = x[6 * n_frames:].reshape(n_points, 3)
    Jc = np.empty((O, 2, 6))
    Jp = np.empty((O, 2, 3))
    c, p = cams[fi], pts[pi]
    for k in range(6):
        d = np.zeros(6)
        d[k] = h * np.maximum(1.0, 1.0)
        Jc[:, :, k] = (project(p, c + d, K) - project(p, c - d, K)) / (2 * d[k])
    for k in range(3):
        d = np.zeros(3)
        d[k] = h
        Jp[:, :, k] = (project(p + d, c, K) - project(p - d, c, K)) / (2 * h)
    return Jc, Jp


# ------------------------------------------------------------------ triangulation (processor.py:246-261)

def triangulate_dlt(P1, P2, x1, x2):
    """Homogeneous two-view DLT for n points — what cv2.triangulatePoints computes
    (processor.py:259; OpenCV calib3d `triangulate.cpp`: rows x*P[2]-P[0], y*P[2]-P[1] for both views,
    4x4, solution = right singular vector of the smallest singular value), then X[:3]/X[3] (:260).
    P1,P2: [n,3,4]; x1,x2: [n,2] -> [n,3].  PARITY UNPINNED (no OpenCV here, no reference fixture)."""
    n = len(x1)
    out = np.empty((n, 3))
    for i in range(n):
        A = np.stack([x1[i, 0] * P1[i, 2] - P1[i, 0], x1[i, 1] * P1[i, 2] - P1[i, 1],
                      x2[i, 0] * P2[i, 2] - P2[i, 0], x2[i, 1] * P2[i, 2] - P2[i, 1]])
        X = np.linalg.svd(A)[2][-1]
        out[i] = X[:3] / X[3]
    return out


# ------------------------------------------------------------------ track bookkeeping (track.py, processor.py)

class Track:
    """Insertion-ordered {frame_ID: (x, y)} with an `updated` flag and a 3-D point (track.py:1-41)."""

    def __init__(self, prev_frame_ID, feature, frame_ID, correspondent):
        self.coordinates = {prev_frame_ID: feature}
        self.coordinates[frame_ID] = correspondent
        self.point = None
        self.updated = False

    def update(self, frame_ID, correspondent):
        self.coordinates[frame_ID] = correspondent
        self.updated = True

    def reset(self):
        self.updated = False

    def wasUpdated(self):
        return self.updated

    def getCoordinate(self, frame_ID):
        return self.coordinates.get(frame_ID)

    def getTriangulationData(self):
        ids = list(self.coordinates)
        return ids[0], ids[-1], self.coordinates[ids[0]], self.coordinates[ids[-1]]

    def getCoordinates(self):
        return self.coordinates

    def setPoint(self, point):
        self.point = point

    def getPoint(self):
        return self.point


def point_tracking(tracks, prev_ID, feature_points, ID, correspondents):
    """Track linking (processor.py:190-243): for each match in order, the FIRST live track whose
    coordinate at prev_ID equals the match's previous-frame point exactly is updated (last writer
    wins on the new coordinate); otherwise a new track is made.  Returns (popped, updated+new)."""
    fresh = []
    for fp, co in zip(feature_points, correspondents):
        key = (fp[0], fp[1])
        val = (co[0], co[1])
        hit = next((t for t in tracks if t.getCoordinate(prev_ID) == key), None)
        if hit is None:
            fresh.append(Track(prev_ID, key, ID, val))
        else:
            hit.update(ID, val)
    kept, popped = [], []
    for t in tracks:
        if t.wasUpdated():
            t.reset()
            kept.append(t)
        else:
            popped.append(t)
    return popped, kept + fresh


def manage_points(tracks):
    """Flatten tracks to BA arrays (processor.py:264-291); return order
    (points, coordinates, frame_indices, point_indices) as at :291."""
    points, coords, fidx, pidx = [], [], [], []
    for i, t in enumerate(tracks):
        points.append(t.getPoint())
        for f, c in t.getCoordinates().items():
            coords.append(c)
            pidx.append(i)
            fidx.append(f)
    return points, coords, fidx, pidx
