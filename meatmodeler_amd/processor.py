"""Drop-in for the hot-path functions of the reference's ``processor.py`` — same names, argument order and
return order, HIP kernels underneath (no CPU fallback: a missing extension or GPU raises).

    reference                                      here
    ---------------------------------------------  ---------------------------------------------------------
    cv2.ORB_create(nfeatures=...)  processor.py:308  ORB_create(nfeatures=...) -> ORB (detectAndCompute on device)
    featureTracking                processor.py:113  featureTracking     (mm_orb_detect_compute + mm_bf_knn2_* + ratio)
    pointTracking                  processor.py:190  pointTracking       (hash join; same list semantics)
    triangulatePoints              processor.py:246  triangulatePoints   (mm_triangulate_dlt, batched over tracks)
    managePoints                   processor.py:264  managePoints
    keyframeTracking               processor.py:61   keyframeTracking    (mm-LK + mm-GFTT: calcOpticalFlowPyrLK / goodFeaturesToTrack)
    increaseContrast               processor.py:12   increaseContrast    (fixed-point L*a*b* + CLAHE), cvtColorBGR2GRAY
    PLY tail                       processor.py:480  savePointCloud      (mm_write_ply)

Video decode, calibration (chessboard / calibrateCamera) and PnP stay outside (SURVEY.md §8 "out of scope").
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import default_context
from .orb_pattern import brief_pattern
from .track import Track


# ----------------------------------------------------------------------------------------------- key points

class KeyPoint:
    """Minimal cv2.KeyPoint look-alike (`.pt`, `.size`, `.angle`, `.response`, `.octave`)."""
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x, y, size, angle, response, octave):
        self.pt = (x, y)
        self.size = size
        self.angle = angle
        self.response = response
        self.octave = octave
        self.class_id = -1


class KeyPoints:
    """Sequence of the key points of one frame.  Holds the device tensors (coordinates, descriptors) so that
    featureTracking never re-uploads them; `kps[i]` builds a `KeyPoint` on demand (its `.pt` is a pair of Python
    floats converted from float32, as cv2.KeyPoint.pt is)."""

    def __init__(self, xy_dev, meta_dev, resp_dev, mom_dev, desc_dev, scales):
        self.xy_dev, self.meta_dev, self.resp_dev, self.mom_dev, self.desc_dev = xy_dev, meta_dev, resp_dev, mom_dev, desc_dev
        self._scales = scales
        self._host = None

    def _h(self):
        if self._host is None:
            self._host = (self.xy_dev.cpu().numpy(), self.meta_dev.cpu().numpy(), self.resp_dev.cpu().numpy(),
                          self.mom_dev.cpu().numpy())
        return self._host

    @property
    def xy(self):
        """[n,2] float32 numpy (level-0 coordinates)."""
        return self._h()[0]

    def __len__(self):
        return self.xy_dev.shape[0]

    def __getitem__(self, i):
        xy, meta, resp, mom = self._h()
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        lvl = int(meta[i, 0])
        ang = math.degrees(math.atan2(float(mom[i, 1]), float(mom[i, 0]))) % 360.0
        return KeyPoint(float(xy[i, 0]), float(xy[i, 1]), 31.0 * float(self._scales[lvl]), ang, float(resp[i]), lvl)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class ORB:
    """Device ORB with the cv2.ORB_create defaults the reference relies on (processor.py:308)."""

    def __init__(self, nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, fastThreshold=20, device=None):
        self.prm = ops.orb_params(nfeatures, nlevels, scaleFactor, edgeThreshold, fastThreshold)
        self.device = device
        self._wsp = {}

    def workspace(self, batch, H, W, device):
        key = (batch, H, W, str(device))
        w = self._wsp.get(key)
        if w is None:
            w = self._wsp[key] = ops.OrbWorkspace(batch, H, W, self.prm, device, brief_pattern())
        return w

    def _as_device_image(self, image):
        if isinstance(image, torch.Tensor):
            img = image
        else:
            import warnings
            with warnings.catch_warnings():          # (a frozen host frame is only read)
                warnings.simplefilter("ignore", UserWarning)
                img = torch.as_tensor(np.ascontiguousarray(image))
        if img.dtype != torch.uint8 or img.dim() != 2:
            raise ValueError("detectAndCompute expects a single-channel uint8 image [H, W]")
        dev = self.device or torch.device("cuda", torch.cuda.current_device())
        img = img.to(dev)
        if img.stride(1) != 1 or img.stride(0) % 4 != 0:
            H, W = img.shape
            pad = torch.zeros((H, (W + 3) // 4 * 4), dtype=torch.uint8, device=dev)
            pad[:, :W] = img
            img = pad[:, :W]
        return img

    def detectAndCompute(self, image, mask=None):
        """-> (KeyPoints, descriptors ndarray [n,32] uint8), like cv2.ORB.detectAndCompute(img, None)."""
        if mask is not None:
            raise NotImplementedError("mask is not supported (the reference always passes None, processor.py:129)")
        img = self._as_device_image(image)
        H, W = img.shape
        wsp = self.workspace(1, H, W, img.device)
        ops.orb_detect_compute(img.as_strided((1, H, W), (H * img.stride(0), img.stride(0), 1)), wsp)
        n = int(wsp.n[0].item())
        kps = KeyPoints(wsp.xy[0, :n].clone(), wsp.meta[0, :n].clone(), wsp.resp[0, :n].clone(), wsp.mom[0, :n].clone(),
                        wsp.desc[0, :n].clone(), ops.orb_level_sizes(H, W, self.prm)[3])
        desc = DeviceBackedDescriptors(kps.desc_dev)
        return kps, desc


class DeviceBackedDescriptors(np.ndarray):
    """ndarray [n,32] uint8 (what cv2 returns) that remembers its device twin."""

    def __new__(cls, desc_dev):
        host = np.asarray(desc_dev.cpu().numpy())
        host.setflags(write=False)          # the device twin stays valid only while the host copy cannot change
        obj = host.view(cls)
        obj._dev = desc_dev
        return obj

    def __array_finalize__(self, obj):
        # views, slices, permutations and copies do NOT inherit the device twin (desc[::-1] or desc[perm] have the
        # same shape but other rows): only the object made in __new__ carries it
        self._dev = None


def ORB_create(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, fastThreshold=20, **_ignored):
    return ORB(nfeatures, scaleFactor, nlevels, edgeThreshold, fastThreshold)


def _device_descriptors(desc, device):
    dev = getattr(desc, "_dev", None)
    if dev is not None and dev.shape[0] == len(desc) and not desc.flags.writeable:
        return dev
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if isinstance(desc, torch.Tensor):
        return desc.to(device).contiguous()
    return torch.as_tensor(np.ascontiguousarray(desc, np.uint8)).to(device)


def _points_xy(points):
    if isinstance(points, KeyPoints):
        return points.xy
    return np.array([p.pt for p in points], np.float32).reshape(-1, 2)


# ----------------------------------------------------------------------------------------------- increaseContrast / grey

def increaseContrast(frame):
    """CLAHE (clip limit 3.5, 8 x 8 tiles) on the L channel of L*a*b* (processor.py:12-26).  frame [H,W,3] u8 BGR
    (ndarray or device tensor) -> ndarray [H,W,3] u8 BGR."""
    ctx = default_context()
    t = frame if isinstance(frame, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(frame, np.uint8))
    out = ops.increase_contrast(t.to(ctx.device).contiguous().unsqueeze(0), 3.5, (8, 8), ctx=ctx)[0]
    return out if isinstance(frame, torch.Tensor) else out.cpu().numpy()


def cvtColorBGR2GRAY(frame):
    """cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) (processor.py:357): [H,W,3] u8 -> [H,W] u8."""
    ctx = default_context()
    t = frame if isinstance(frame, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(frame, np.uint8))
    out = ops.bgr_to_grey(t.to(ctx.device).contiguous(), ctx)
    return out if isinstance(frame, torch.Tensor) else out.cpu().numpy()


# ----------------------------------------------------------------------------------------------- keyframeTracking

def _upload_u8(a, device):
    """host u8 array (possibly frozen: torch warns about wrapping read-only memory, which is only read here) -> device."""
    import warnings
    a = np.ascontiguousarray(a, np.uint8)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        t = torch.from_numpy(a)
    return t.to(device)


_PYR_CACHE = []          # [(weakref to the host frame, max_level, device pyramid)]: the frame of one call is `prev` of the next


def _frame_pyramid(frame_grey, max_level, ctx):
    import weakref
    if isinstance(frame_grey, torch.Tensor):
        return ops.pyramid(frame_grey.to(ctx.device).contiguous(), max_level, ctx)
    for ref, lv, pyr in _PYR_CACHE:
        if ref() is frame_grey and lv == max_level and not frame_grey.flags.writeable and frame_grey.base is None:
            return pyr
    img = _upload_u8(frame_grey, ctx.device)
    pyr = ops.pyramid(img, max_level, ctx)
    try:
        # (only frames that cannot change behind the cache's back are remembered: the caller opts in by freezing an array
        # that OWNS its data -- a read-only view of a writable base can still change through the base)
        if not frame_grey.flags.writeable and frame_grey.base is None:
            _PYR_CACHE.append((weakref.ref(frame_grey), max_level, pyr))
            del _PYR_CACHE[:-2]
    except TypeError:
        pass
    return pyr


def calcOpticalFlowPyrLK(prevImg, nextImg, prevPts, nextPts=None, winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01),
                         **_ignored):
    """cv2.calcOpticalFlowPyrLK (processor.py:79) -> (nextPts [n,1,2] f32, status [n,1] u8, err [n,1] f32).
    prevPts = None raises, as the cv2 call does (the reference reaches it when goodFeaturesToTrack found no corner at the
    last keyframe, processor.py:104-110, and fails there -- it must not turn into "never a keyframe again" here);
    an EMPTY point array gives (None, None, None), cv2's empty outputs."""
    if prevPts is None:
        raise ValueError("calcOpticalFlowPyrLK: prevPts is None (no points to track; cv2 raises "
                         "'(npoints = prevPtsMat.checkVector(2, CV_32F, true)) >= 0' here)")
    if len(prevPts) == 0:
        return None, None, None
    ctx = default_context()
    pp = _frame_pyramid(prevImg, maxLevel, ctx)
    pn = _frame_pyramid(nextImg, maxLevel, ctx)
    pts = torch.as_tensor(np.ascontiguousarray(np.asarray(prevPts, np.float32).reshape(-1, 2))).to(ctx.device)
    max_count = criteria[1] if len(criteria) > 1 else 30
    eps = criteria[2] if len(criteria) > 2 else 0.01
    nx, st, err = ops.lk_track(pp, pn, pts, winSize, max_count, eps, ctx)
    return (nx.cpu().numpy().reshape(-1, 1, 2), st.cpu().numpy().reshape(-1, 1), err.cpu().numpy().reshape(-1, 1))


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, mask=None, blockSize=3, useHarrisDetector=False,
                        k=0.04, **_ignored):
    """cv2.goodFeaturesToTrack (processor.py:104) -> [n,1,2] f32 or None without corners."""
    if mask is not None or useHarrisDetector:
        raise NotImplementedError("mask / Harris detector are not supported (the reference passes mask=None, processor.py:105)")
    ctx = default_context()
    img = image.to(ctx.device) if isinstance(image, torch.Tensor) else _upload_u8(image, ctx.device)
    c = ops.good_features(img.contiguous(), maxCorners, qualityLevel, minDistance, blockSize, ctx)
    return c.reshape(-1, 1, 2) if len(c) else None


def keyframeTracking(frame_grey, prev_frame_grey, prev_frame_points, accumulated_error, lk_params, feature_params,
                     threshold=0.2):
    """Is this frame a keyframe?  (processor.py:61-110.)  Tracks the previous frame's points into this frame with
    pyramidal LK, adds the mean tracking error to `accumulated_error`, and once that exceeds threshold * width declares
    a keyframe, resets the error and re-seeds the points with goodFeaturesToTrack.
    -> (is_keyframe, new prev_frame_grey, new prev_frame_points, new accumulated_error)."""
    p, st, err = calcOpticalFlowPyrLK(prev_frame_grey, frame_grey, prev_frame_points, None, **lk_params)
    if p is not None:
        good_new = p[st == 1]
        prev_frame_grey = frame_grey
        prev_frame_points = good_new.reshape(-1, 1, 2)
        if err is not None:
            new_err = np.nan_to_num(err)
            new_err[new_err < 0] = 0
            accumulated_error += np.average(new_err)
        if accumulated_error > threshold * frame_grey.shape[1]:
            accumulated_error = 0
            prev_frame_points = goodFeaturesToTrack(prev_frame_grey, mask=None, **feature_params)
            return True, prev_frame_grey, prev_frame_points, accumulated_error
    return False, prev_frame_grey, prev_frame_points, accumulated_error


# ----------------------------------------------------------------------------------------------- PLY export

def savePointCloud(points, path):
    """The reference's export (processor.py:480-485: PyntCloud(DataFrame(points, columns x, y, z)).to_file(path +
    "Cloud.ply")): binary little-endian PLY with double x, y, z.  -> the file name."""
    from ._lib import lib as _l, MMError as _E
    pts = np.ascontiguousarray(np.asarray(points, np.float64).reshape(-1, 3))
    filename = path + "Cloud.ply"
    rc = _l.mm_write_ply(filename.encode(), pts.ctypes.data_as(__import__("ctypes").c_void_p), len(pts))
    if rc != 0:
        raise _E(f"mm_write_ply({filename!r}) failed ({rc})")
    return filename


# ----------------------------------------------------------------------------------------------- featureTracking

def featureTracking(new_keyframe, prev_orb_points, prev_orb_descriptors, orb, flann_params, threshold=0.75):
    """Detect + describe the new keyframe and match the previous keyframe against it (processor.py:113-142).

    `flann_params` is accepted for signature compatibility and ignored: the matcher is the exact brute-force
    Hamming 2-NN that FLANN-LSH approximates (query = previous keyframe, train = new keyframe, processor.py:133);
    a match is kept iff it has two neighbours and d0 < threshold * d1 (processor.py:136-137).
    Returns (prev_matches [M,2] f64, curr_matches [M,2] f64, new_points, new_descriptors); M = 0 gives the
    reference's `np.array([])`.
    """
    new_points, new_descriptors = orb.detectAndCompute(new_keyframe, None)
    t_dev = _device_descriptors(new_descriptors, None)
    q_dev = _device_descriptors(prev_orb_descriptors, t_dev.device)
    idx, dist = ops.bf_knn2(q_dev, t_dev)
    pairs, m = ops.ratio_filter_batched(idx.unsqueeze(0), dist.unsqueeze(0), threshold)
    M = int(m[0].item())
    good = pairs[0, :M].cpu().numpy()
    if M == 0:
        return np.array([]), np.array([]), new_points, new_descriptors
    pxy = _points_xy(prev_orb_points)
    nxy = _points_xy(new_points)
    prev_matches = pxy[good[:, 0]].astype(np.float64)
    curr_matches = nxy[good[:, 1]].astype(np.float64)
    return prev_matches, curr_matches, new_points, new_descriptors


# ----------------------------------------------------------------------------------------------- pointTracking

def pointTracking(tracks, prev_keyframe_ID, feature_points, keyframe_ID, correspondents):
    """Track linking with the reference's list semantics (processor.py:190-243) as a hash join:
    the FIRST live track whose coordinate at `prev_keyframe_ID` equals the match's previous-frame point is
    updated (a later match on the same track overwrites its new coordinate); other matches start new tracks,
    appended after the surviving ones in match order; tracks not updated are popped in list order."""
    first = {}
    for pos, track in enumerate(tracks):
        c = track.getCoordinate(prev_keyframe_ID)
        if c is not None:
            first.setdefault((float(c[0]), float(c[1])), pos)
    new_tracks = []
    for fp, co in zip(feature_points, correspondents):
        feature_point = (fp[0], fp[1])
        correspondent = (co[0], co[1])
        pos = first.get((float(fp[0]), float(fp[1])))
        if pos is None:
            new_tracks.append(Track(prev_keyframe_ID, feature_point, keyframe_ID, correspondent))
        else:
            tracks[pos].update(keyframe_ID, correspondent)
    updated_tracks, popped_tracks = [], []
    for track in tracks:
        if track.wasUpdated():
            track.reset()
            updated_tracks.append(track)
        else:
            popped_tracks.append(track)
    updated_tracks += new_tracks
    return popped_tracks, updated_tracks


# ----------------------------------------------------------------------------------------------- triangulatePoints

def triangulatePoints(tracks, projections):
    """Two-view DLT of every track from its first and last observation (processor.py:246-261), one kernel launch
    for all tracks; each track receives a (1,3) float64 array via setPoint, as in the reference."""
    n = len(tracks)
    if n == 0:
        return
    f0 = np.empty(n, np.int32)
    f1 = np.empty(n, np.int32)
    x0 = np.empty((n, 2), np.float64)
    x1 = np.empty((n, 2), np.float64)
    for i, track in enumerate(tracks):
        a, b, pa, pb = track.getTriangulationData()
        f0[i], f1[i] = a, b
        x0[i] = (pa[0], pa[1])
        x1[i] = (pb[0], pb[1])
    n_proj = len(projections)
    if min(f0.min(), f1.min()) < -n_proj or max(f0.max(), f1.max()) >= n_proj:
        raise IndexError("list index out of range")          # what projections[frame_ID] raises, processor.py:257-258
    f0[f0 < 0] += n_proj                                       # (a negative ID indexes from the end, as in the reference)
    f1[f1 < 0] += n_proj
    ctx = default_context()
    dev = ctx.device
    proj = torch.as_tensor(np.ascontiguousarray(np.asarray(projections, np.float64).reshape(-1, 3, 4))).to(dev)
    X = ops.triangulate_dlt(proj, torch.as_tensor(f0).to(dev), torch.as_tensor(f1).to(dev),
                            torch.as_tensor(x0).to(dev), torch.as_tensor(x1).to(dev), ctx).cpu().numpy()
    for i, track in enumerate(tracks):
        track.setPoint(X[i:i + 1].copy())


# ----------------------------------------------------------------------------------------------- managePoints

def managePoints(tracks):
    """Flatten tracks into BA arrays (processor.py:264-291).  Return order is the reference's:
    (points, coordinates, frame_indices, point_indices)."""
    points, coordinates, frame_indices, point_indices = [], [], [], []
    for point_index, track in enumerate(tracks):
        points.append(track.getPoint())
        obs = track.getCoordinates()
        coordinates.extend(obs.values())
        frame_indices.extend(obs.keys())
        point_indices.extend([point_index] * len(obs))
    return points, coordinates, frame_indices, point_indices


# ----------------------------------------------------------------------------------------------- the driver loop

def processFrames(frames, camera_matrix, extrinsic_for_frame, path, lk_params, feature_params, flann_params=None,
                  threshold=0.1, nfeatures=20000, adjust=True):
    """The body of the reference's `process` (processor.py:294-489) on frames that are already decoded, in the
    reference's call order: contrast -> grey -> keyframe gate (:357-365) -> on a keyframe ORB + matching + track linking
    (:378-391) -> after the last frame triangulation, flattening, bundle adjustment and the PLY export (:418-485).
    What stays outside (video decode, chessboard detection, calibrate, poseEstimation, adjustPose)
    is supplied by the caller: `frames` = iterable of [H,W,3] u8 BGR images, `camera_matrix` = K, and
    `extrinsic_for_frame(i) -> 3x4` for the keyframes (every keyframe counts as "has a chessboard").
    -> dict(points [P,3], extrinsics (list of 4x4) | None, keyframes (frame indices), tracks, file)."""
    from . import bundleAdjuster
    orb = ORB_create(nfeatures=nfeatures)
    it = iter(frames)
    start_frame = next(it)
    prev_frame_grey = cvtColorBGR2GRAY(increaseContrast(start_frame))
    prev_frame_grey.setflags(write=False)          # (lets the LK pyramid of a frame be reused by the next call)
    prev_frame_points = goodFeaturesToTrack(prev_frame_grey, mask=None, **feature_params)
    accumulative_error = 0
    prev_orb_points, prev_orb_descriptors = orb.detectAndCompute(prev_frame_grey, None)
    keyframes = [0]
    tracks, popped_tracks = [], []
    prev_keyframe_ID, keyframe_ID = 0, 1
    for index, frame in enumerate(it, start=1):
        frame_grey = cvtColorBGR2GRAY(increaseContrast(frame))
        frame_grey.setflags(write=False)
        is_keyframe, prev_frame_grey, prev_frame_points, accumulative_error = keyframeTracking(
            frame_grey, prev_frame_grey, prev_frame_points, accumulative_error, lk_params, feature_params,
            threshold=threshold)
        if is_keyframe:
            keyframes.append(index)
            prev_matches, curr_matches, prev_orb_points, prev_orb_descriptors = featureTracking(
                frame_grey, prev_orb_points, prev_orb_descriptors, orb, flann_params)
            new_popped_tracks, tracks = pointTracking(tracks, prev_keyframe_ID, prev_matches, keyframe_ID, curr_matches)
            popped_tracks += new_popped_tracks
            prev_keyframe_ID = keyframe_ID
            keyframe_ID += 1
    popped_tracks += tracks
    out = dict(points=np.zeros((0, 3)), extrinsics=None, keyframes=keyframes, tracks=popped_tracks, file=None)
    if not popped_tracks:
        return out
    K = np.asarray(camera_matrix, float)
    extrinsic_matrices = [np.asarray(extrinsic_for_frame(i), float)[:3, :] for i in keyframes]
    projections = [K @ e for e in extrinsic_matrices]                     # processor.py:448
    triangulatePoints(popped_tracks, projections)
    points, points_2d, frame_indices, point_indices = managePoints(popped_tracks)
    if adjust:
        adjusted_points, adjusted_positions = bundleAdjuster.adjustPoints(
            np.array(extrinsic_matrices), K, np.array(points), np.array(points_2d), np.array(frame_indices),
            np.array(point_indices))
        out.update(points=adjusted_points, extrinsics=adjusted_positions)
    else:
        out.update(points=np.array(points).reshape(-1, 3))
    if path is not None:
        out["file"] = savePointCloud(out["points"], path)
    return out
