"""Thin tensor-level wrappers over the C ABI (include/meatmodeler.h).

torch is used for device memory and streams only; every computation below is a HIP kernel launched through
`libmeatmodeler_hip.so`.  All functions take/return torch tensors on the context's device.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr, vp, OrbParams, BAProblem, default_context


def _i32(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return t


# ------------------------------------------------------------------------------------------------ matching

def bf_knn2_batched(q, t, nq=None, nt=None, ctx=None):
    """q [n_pairs, nq_cap, 32] u8, t [n_pairs, nt_cap, 32] u8 (may be strided views over a descriptor table
    as long as rows are contiguous); nq / nt optional int32 [n_pairs] valid counts.
    Returns idx, dist int32 [n_pairs, nq_cap, 2]."""
    ctx = ctx or default_context()
    assert q.dtype == torch.uint8 and t.dtype == torch.uint8 and q.shape[-1] == 32 and t.shape[-1] == 32
    assert q.stride(-1) == 1 and q.stride(-2) == 32 and t.stride(-1) == 1 and t.stride(-2) == 32
    n_pairs, nq_cap, nt_cap = q.shape[0], q.shape[1], t.shape[1]
    assert t.shape[0] == n_pairs
    both = torch.full((2, n_pairs, nq_cap, 2), -1, dtype=torch.int32, device=q.device)      # (one fill, two views)
    idx, dist = both[0], both[1]
    if n_pairs == 0 or nq_cap == 0:
        return idx, dist
    wsb = lib.mm_bf_workspace_bytes(n_pairs, nq_cap, nt_cap)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    qs = q.stride(0) if n_pairs > 1 else 0
    ts = t.stride(0) if n_pairs > 1 else 0
    ctx.check(lib.mm_bf_knn2_batched(ctx.h, ptr(q), ptr(nq), nq_cap, qs, ptr(t), ptr(nt), nt_cap, ts, n_pairs,
                                     ptr(idx), ptr(dist), ptr(ws), wsb), "mm_bf_knn2_batched")
    return idx, dist


def bf_knn2(q, t, ctx=None):
    """Single pair: q [nq,32], t [nt,32] -> idx, dist int32 [nq,2] (mm_bf_knn2_hamming)."""
    ctx = ctx or default_context()
    assert q.dtype == torch.uint8 and t.dtype == torch.uint8 and q.is_contiguous() and t.is_contiguous()
    nq, nt = q.shape[0], t.shape[0]
    idx = torch.full((nq, 2), -1, dtype=torch.int32, device=q.device)
    dist = torch.full((nq, 2), -1, dtype=torch.int32, device=q.device)
    if nq == 0:
        return idx, dist
    wsb = lib.mm_bf_workspace_bytes(1, nq, nt)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    ctx.check(lib.mm_bf_knn2_hamming(ctx.h, ptr(q), nq, ptr(t), nt, ptr(idx), ptr(dist), ptr(ws), wsb),
              "mm_bf_knn2_hamming")
    return idx, dist


def ratio_filter_batched(idx, dist, threshold=0.75, nq=None, ctx=None):
    """idx/dist [n_pairs, nq_cap, 2] -> pairs int32 [n_pairs, nq_cap, 2] (queryIdx, trainIdx), m int32 [n_pairs]."""
    ctx = ctx or default_context()
    n_pairs, nq_cap = idx.shape[0], idx.shape[1]
    pairs = torch.full((n_pairs, nq_cap, 2), -1, dtype=torch.int32, device=idx.device)
    m = torch.zeros(n_pairs, dtype=torch.int32, device=idx.device)
    if n_pairs == 0:
        return pairs, m
    ctx.check(lib.mm_ratio_filter_batched(ctx.h, ptr(_i32(idx)), ptr(_i32(dist)), ptr(nq), nq_cap, n_pairs,
                                          float(threshold), ptr(pairs), ptr(m)), "mm_ratio_filter_batched")
    return pairs, m


# ------------------------------------------------------------------------------------------------ ORB

def orb_params(nfeatures, nlevels=8, scale_factor=1.2, edge_threshold=31, fast_threshold=20):
    return OrbParams(int(nfeatures), int(nlevels), int(edge_threshold), int(fast_threshold), float(scale_factor), 0)


def orb_level_sizes(H, W, prm):
    L = prm.nlevels
    w = (C.c_int32 * L)()
    h = (C.c_int32 * L)()
    n = (C.c_int32 * L)()
    s = (C.c_float * L)()
    rc = lib.mm_orb_level_sizes(H, W, C.byref(prm), w, h, n, s)
    if rc != 0:
        raise _lib.MMError(f"mm_orb_level_sizes failed ({rc})")
    return np.array(w), np.array(h), np.array(n), np.array(s, np.float32)


class OrbWorkspace:
    """Reusable device workspace + output buffers for a fixed (batch, H, W, params)."""

    def __init__(self, batch, H, W, prm, device, pattern):
        self.batch, self.H, self.W, self.prm = batch, H, W, prm
        nb = lib.mm_orb_workspace_bytes(batch, H, W, C.byref(prm))
        if nb == 0:
            raise _lib.MMError("mm_orb_workspace_bytes: unsupported geometry")
        self.nbytes = nb
        self.ws = torch.empty(nb, dtype=torch.uint8, device=device)
        self.pattern = torch.as_tensor(np.ascontiguousarray(pattern, np.int8)).to(device)
        cap = prm.nfeatures
        self.xy = torch.zeros((batch, cap, 2), dtype=torch.float32, device=device)
        self.meta = torch.zeros((batch, cap, 4), dtype=torch.int32, device=device)
        self.resp = torch.zeros((batch, cap), dtype=torch.float32, device=device)
        self.mom = torch.zeros((batch, cap, 2), dtype=torch.int32, device=device)
        self.desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device=device)
        self.n = torch.zeros(batch, dtype=torch.int32, device=device)


def orb_detect_compute(imgs, wsp, ctx=None, out=None):
    """imgs [B,H,W] u8 (B <= wsp.batch).  Fills (and returns) wsp.xy/meta/resp/mom/desc/n for the first B frames,
    or the tensors in `out` = (xy, meta, resp, mom, desc, n) if given (each with leading dim >= B)."""
    ctx = ctx or default_context()
    assert imgs.dtype == torch.uint8 and imgs.dim() == 3 and imgs.stride(2) == 1
    B, H, W = imgs.shape
    assert B <= wsp.batch and H == wsp.H and W == wsp.W and imgs.stride(0) == H * imgs.stride(1)
    xy, meta, resp, mom, desc, n = out if out is not None else (wsp.xy, wsp.meta, wsp.resp, wsp.mom, wsp.desc, wsp.n)
    ctx.check(lib.mm_orb_detect_compute(ctx.h, ptr(imgs), B, H, W, imgs.stride(1), C.byref(wsp.prm), ptr(wsp.pattern),
                                        ptr(wsp.ws), wsp.nbytes, ptr(xy), ptr(meta), ptr(resp), ptr(mom), ptr(desc),
                                        ptr(n)), "mm_orb_detect_compute")
    return xy, meta, resp, mom, desc, n


# ------------------------------------------------------------------------------------------------ triangulation

def triangulate_dlt(proj, f0, f1, x0, x1, ctx=None):
    """proj [F,3,4] f64, f0/f1 int32 [n], x0/x1 f64 [n,2] -> X f64 [n,3]."""
    ctx = ctx or default_context()
    n = f0.shape[0]
    X = torch.empty((n, 3), dtype=torch.float64, device=proj.device)
    if n:
        lim = torch.stack([f0.min(), f1.min(), f0.max(), f1.max()]).tolist()
        if min(lim[:2]) < 0 or max(lim[2:]) >= proj.shape[0]:
            raise IndexError("triangulate_dlt: frame index outside the projection table")
        ctx.check(lib.mm_triangulate_dlt(ctx.h, ptr(proj.contiguous()), ptr(_i32(f0)), ptr(_i32(f1)),
                                         ptr(x0.contiguous()), ptr(x1.contiguous()), n, ptr(X)), "mm_triangulate_dlt")
    return X


# ------------------------------------------------------------------------------------------------ track linking

def link_tracks_device(kp_count, kp_xy, match_count, matches, ctx=None):
    """pointTracking over a whole clip on the device (mm_link_tracks_device; reference processor.py:190-243,418).
    kp_count [F] i32, kp_xy [F,cap,2] f32, match_count [F-1] i32, matches [F-1,cap,2] i32 (device tensors)
    -> (track_ptr [T+1] i32, obs_frame [O] i32, obs_kp [O] i32, bad) device tensors in the reference's final track
    order.  One host read-back (the two counts) sizes the views."""
    ctx = ctx or default_context()
    F, cap = kp_xy.shape[0], kp_xy.shape[1]
    d = kp_xy.device
    npairs = max(F - 1, 0)
    if matches.shape[0] != npairs or (npairs and matches.shape[1] != cap) or match_count.shape[0] != npairs:
        raise ValueError("link_tracks_device: matches must be [F-1, cap, 2] with match_count [F-1]")
    track_ptr = torch.empty(npairs * cap + 1, dtype=torch.int32, device=d)
    obs_frame = torch.empty(max(2 * npairs * cap, 1), dtype=torch.int32, device=d)
    obs_kp = torch.empty_like(obs_frame)
    counts = torch.zeros(3, dtype=torch.int64, device=d)
    wsb = lib.mm_link_workspace_bytes(F, cap)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=d)
    ctx.check(lib.mm_link_tracks_device(ctx.h, F, cap, ptr(_i32(kp_count)), ptr(kp_xy.contiguous()),
                                        ptr(_i32(match_count)), ptr(_i32(matches)), ptr(ws), ws.numel(), ptr(track_ptr),
                                        ptr(obs_frame), ptr(obs_kp), ptr(counts)), "mm_link_tracks_device")
    nt, no, bad = (int(v) for v in counts.cpu())
    return track_ptr[:nt + 1], obs_frame[:no], obs_kp[:no], bool(bad)


def flatten_tracks(track_ptr, obs_frame, obs_kp, kp_xy, sel=None, t_lo=0, n_sel=None, frame_offset=0, ctx=None):
    """managePoints on the device (mm_flatten_tracks; reference processor.py:264-291): the flat observation arrays of a
    selection of tracks.  track_ptr [T+1] i32, obs_frame / obs_kp [O] i32, kp_xy [F,cap,2] f32 (device tensors).
    sel = None: the tracks t_lo .. t_lo + n_sel - 1 (default: all); sel [n] integer tensor: these tracks, in this order.
    -> (coords [n_obs,2] f64, frame_indices [n_obs] i32 (minus frame_offset), point_indices [n_obs] i32) device tensors."""
    ctx = ctx or default_context()
    d = kp_xy.device
    cap = kp_xy.shape[1]
    tp = track_ptr.to(torch.int32).contiguous()
    T = tp.numel() - 1
    out_ptr = None
    if sel is None:
        n_sel = T - t_lo if n_sel is None else int(n_sel)
        if n_sel < 0 or t_lo < 0 or t_lo + n_sel > T:
            raise ValueError("flatten_tracks: range outside the track list")
        n_obs = int((tp[t_lo + n_sel] - tp[t_lo]).item()) if n_sel else 0      # (one read-back sizes the outputs)
        sel_t = None
    else:
        sel_t = sel.to(torch.int32).contiguous()
        n_sel = sel_t.numel()
        out_ptr = torch.empty(n_sel + 1, dtype=torch.int64, device=d)
        ctx.check(lib.mm_flatten_offsets(ctx.h, ptr(tp), ptr(sel_t) if n_sel else None, n_sel, ptr(out_ptr)), "mm_flatten_offsets")
        n_obs = int(out_ptr[n_sel].item())
    coords = torch.empty((n_obs, 2), dtype=torch.float64, device=d)
    fi = torch.empty(n_obs, dtype=torch.int32, device=d)
    pi = torch.empty(n_obs, dtype=torch.int32, device=d)
    if n_obs:
        ctx.check(lib.mm_flatten_tracks(ctx.h, ptr(tp), ptr(obs_frame.to(torch.int32).contiguous()), ptr(obs_kp.to(torch.int32).contiguous()),
                                        ptr(kp_xy.contiguous()), cap,
                                        ptr(sel_t) if sel_t is not None else None, int(t_lo), n_sel,
                                        ptr(out_ptr) if out_ptr is not None else None, n_obs, int(frame_offset), ptr(coords),
                                        ptr(fi), ptr(pi)), "mm_flatten_tracks")
    return coords, fi, pi


# ------------------------------------------------------------------------------------------------ bundle adjustment

def ba_build_index(F, P, fi, pi):
    """Host CSR build (mm_ba_build_index).  fi, pi: int32 numpy [O]."""
    fi = np.ascontiguousarray(fi, np.int32)
    pi = np.ascontiguousarray(pi, np.int32)
    O = fi.size
    pt_ptr = np.zeros(P + 1, np.int32)
    cam_ptr = np.zeros(F + 1, np.int32)
    pt_obs = np.zeros(max(O, 1), np.int32)
    cam_obs = np.zeros(max(O, 1), np.int32)
    i32p = _lib.c_i32p
    rc = lib.mm_ba_build_index(F, P, O, fi.ctypes.data_as(i32p), pi.ctypes.data_as(i32p), pt_ptr.ctypes.data_as(i32p),
                               pt_obs.ctypes.data_as(i32p), cam_ptr.ctypes.data_as(i32p), cam_obs.ctypes.data_as(i32p))
    if rc != 0:
        raise ValueError(f"mm_ba_build_index failed ({rc}): frame/point index out of range")
    return pt_ptr, pt_obs[:O], cam_ptr, cam_obs[:O]


class BADevice:
    """Device-resident BA problem: observation arrays, CSR indices and the mm_ba_problem descriptor."""

    def __init__(self, K, fi, pi, obs, F, P, device, ctx=None, pairs=True, max_band_span=192):
        """fi, pi: int [O] (numpy or device tensors), obs [O,2] f64 (numpy or device tensor).  All index structures
        (CSR by point, CSR by camera, co-observation pair list and its chunk table) are built on the device: torch
        provides the sorts / scans (index plumbing), two HIP kernels enumerate the pairs."""
        self.ctx = ctx or default_context()
        self._mdot = None
        dev = device
        self.device = dev
        self.F, self.P, self.O = int(F), int(P), int(len(fi))
        F, P, O = self.F, self.P, self.O

        def to_dev(a, dtype):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dtype).contiguous()
            return torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dtype).contiguous()

        self.K = to_dev(np.asarray(K, np.float64).reshape(9), torch.float64)
        self.fi = to_dev(fi, torch.int32).reshape(-1)
        self.pi = to_dev(pi, torch.int32).reshape(-1)
        self.obs = to_dev(obs, torch.float64).reshape(-1, 2)
        i64 = dict(dtype=torch.int64, device=dev)
        if O:
            lim = torch.stack([self.fi.min(), self.fi.max(), self.pi.min(), self.pi.max(),
                               (self.pi[1:] >= self.pi[:-1]).all().to(torch.int32)]).tolist()
            if lim[0] < 0 or lim[1] >= F or lim[2] < 0 or lim[3] >= P:
                raise ValueError("frame / point index out of range")
            point_major = bool(lim[4])
        else:
            point_major = True
        # CSR by point (identity permutation when the observations are already point-major, as managePoints emits them)
        if point_major:
            self.pt_obs = torch.arange(O, dtype=torch.int32, device=dev)
        else:
            self.pt_obs = torch.sort(self.pi, stable=True)[1].to(torch.int32)
        self.pt_ptr = torch.zeros(P + 1, dtype=torch.int32, device=dev)
        self.cam_ptr = torch.zeros(F + 1, dtype=torch.int32, device=dev)
        if O:
            self.pt_ptr[1:] = torch.cumsum(torch.bincount(self.pi, minlength=P), 0).to(torch.int32)
            self.cam_ptr[1:] = torch.cumsum(torch.bincount(self.fi, minlength=F), 0).to(torch.int32)
            self.cam_obs = torch.sort(self.fi, stable=True)[1].to(torch.int32)      # stable: observation order kept
        else:
            self.cam_obs = torch.zeros(0, dtype=torch.int32, device=dev)
        self.cam_span = 0
        self.max_band_span = int(max_band_span)
        self.slabs = None
        # build + solve overlapped on two streams (mm_ba_schur_solve).  Needs concurrent kernel execution: tools that
        # serialise kernels (rocprofv3 --pmc, launch-blocking debug modes) make the consumer wait for a producer that
        # cannot start; its bounded spins then give up (info = -1) and the driver falls back to one after the other.
        # Off by default since the factorisation became two-ended: its 137 workgroups use the whole register file of
        # their CUs (one wave per SIMD), the build slows down by 1.6x beside them and build-then-solve (0.32 + 0.59 ms at
        # 500 cameras) beats the overlapped pair (0.93 ms).  MM_SCHUR_OVERLAP=1 selects the overlapped path.
        self.overlap = os.environ.get("MM_SCHUR_OVERLAP", "0") != "0"
        self.pb = BAProblem(F, P, O, ptr(self.K), ptr(self.fi), ptr(self.pi), ptr(self.obs),
                            ptr(self.pt_ptr), ptr(self.pt_obs), ptr(self.cam_ptr), ptr(self.cam_obs),
                            0, 0, 0, None, None, 0, None, None, None, None, None, None)
        self.n_pairs = 0
        if O:
            # widest camera span of any point: cameras further apart never share a point, so the reduced camera system
            # is block banded with this half-width (tracks from consecutive-keyframe matching span a few frames only)
            cnt = torch.empty(O, dtype=torch.int32, device=dev)
            span_t = torch.zeros(1, dtype=torch.int32, device=dev)
            self.ctx.check(lib.mm_ba_pairs_count(self.ctx.h, C.byref(self.pb), ptr(cnt), ptr(span_t)),
                           "mm_ba_pairs_count")
            offs = torch.cumsum(cnt, 0, dtype=torch.int64)
            span, n = torch.stack([span_t[0].to(torch.int64), offs[-1]]).tolist()
            self.cam_span = int(span)
            self.pb.cam_span = self.cam_span
            # banded problems: co-observation pair list for the atomic-free, bitwise reproducible Schur kernel;
            # wide spans keep the general kernel
            if pairs and self.cam_span <= max_band_span and self.cam_span < F and F * (self.cam_span + 1) < 2 ** 31:
                key = torch.empty(n, dtype=torch.int32, device=dev)
                po = torch.empty(n, dtype=torch.int32, device=dev)
                po2 = torch.empty(n, dtype=torch.int32, device=dev)
                offs_ex = (offs - cnt).contiguous()
                self.ctx.check(lib.mm_ba_pairs_emit(self.ctx.h, C.byref(self.pb), ptr(offs_ex), self.cam_span, ptr(key),
                                                    ptr(po), ptr(po2)), "mm_ba_pairs_emit")
                key_s, perm = torch.sort(key, stable=True)          # fixed order inside every segment
                self.pair_o, self.pair_o2 = po[perm].contiguous(), po2[perm].contiguous()
                seg_ids, counts = torch.unique_consecutive(key_s, return_counts=True)
                seg_hi = torch.cumsum(counts, 0)
                seg_lo = seg_hi - counts
                # pairs per chunk = per wave of the pair kernel (a multiple of 64: whole lane strides).  512 since the second
                # half of round 4: 230 us per build at the bench shape against 249 with 256 (384: 230; 1024: slower)
                CH = max(64, int(os.environ.get("MM_SCHUR_CHUNK", "512")) // 64 * 64)
                cntc = (counts + (CH - 1)) // CH
                first_hi = torch.cumsum(cntc, 0)
                first = first_hi - cntc
                nchunks = int(first_hi[-1].item())
                chunk_seg = torch.repeat_interleave(torch.arange(len(seg_ids), **i64), cntc, output_size=nchunks)
                chunk_begin = seg_lo[chunk_seg] + CH * (torch.arange(nchunks, **i64) - first[chunk_seg])
                chunk_end = torch.minimum(chunk_begin + CH, seg_hi[chunk_seg])
                i32 = lambda t: t.to(torch.int32).contiguous()
                self.seg_ids = i32(seg_ids)
                self.seg_chunk_ptr = i32(torch.cat([first, first_hi[-1:]]))
                self.chunk_seg, self.chunk_begin, self.chunk_end = i32(chunk_seg), i32(chunk_begin), i32(chunk_end)
                self.n_pairs = int(n)
                self.pb.n_seg = len(seg_ids)
                self.pb.n_chunks = nchunks
                self.pb.seg_ids, self.pb.seg_chunk_ptr = ptr(self.seg_ids), ptr(self.seg_chunk_ptr)
                self.pb.chunk_seg, self.pb.chunk_begin, self.pb.chunk_end = ptr(self.chunk_seg), ptr(self.chunk_begin), ptr(self.chunk_end)
                self.pb.pair_o, self.pb.pair_o2 = ptr(self.pair_o), ptr(self.pair_o2)
                self.pair_p = self.pi[self.pair_o.long()].contiguous()   # saves the pair kernel a dependent gather
                self.pb.pair_p = ptr(self.pair_p)
                # camera slabs for the overlapped build + solve (mm_ba_schur_solve): first segment / chunk of each slab
                n_slabs = 8
                cps = -(-F // n_slabs)
                if cps >= 16:
                    seg_cam = seg_ids.to(torch.int64) // (self.cam_span + 1)
                    bounds = torch.arange(n_slabs + 1, **i64) * cps
                    sseg = torch.searchsorted(seg_cam, bounds)
                    schunk = torch.cat([first, first_hi[-1:]])[sseg]
                    self.slabs = (n_slabs, cps, np.ascontiguousarray(sseg.cpu().numpy(), np.int64),
                                  np.ascontiguousarray(schunk.cpu().numpy(), np.int64))
        self._ws = torch.empty(2048 * 8, dtype=torch.uint8, device=dev)
        self._cost2 = torch.zeros(1, dtype=torch.float64, device=dev)
        self._S = None  # reduced camera system, allocated once (6F x 6F doubles)
        self._schur_ws = None
        self._chol_ws = None

    def residual(self, cams, pts, want_res=False, cost_out=None):
        """-> (sum of squared residuals as a 1-element device tensor, res [O,2] or None).  cost_out: where to put the
        sum (a 1-element f64 device tensor, e.g. a slot of the trust-region driver's scalar board)."""
        res = torch.empty((self.O, 2), dtype=torch.float64, device=self.device) if want_res else None
        cost2 = cost_out if cost_out is not None else torch.empty(1, dtype=torch.float64, device=self.device)
        self.ctx.check(lib.mm_ba_residual(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(res), ptr(cost2),
                                          ptr(self._ws), self._ws.numel()), "mm_ba_residual")
        return cost2, res

    def jacobian(self, cams, pts):
        Jc = torch.empty((self.O, 2, 6), dtype=torch.float64, device=self.device)
        Jp = torch.empty((self.O, 2, 3), dtype=torch.float64, device=self.device)
        self.ctx.check(lib.mm_ba_jacobian(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(Jc), ptr(Jp)),
                       "mm_ba_jacobian")
        return Jc, Jp

    def normal_eq(self, cams, pts, want_cams=True, want_pts=True, out=None):
        """-> (B [F,6,6], gc [F,6], C [P,6], gp [P,3]); `out` = the four tensors to fill (any contiguous storage of
        the right size, e.g. the two halves of the flat gradient vector for gc / gp)."""
        d = self.device
        if out is not None:
            B, gc, Cb, gp = out
            for t_ in out:
                assert t_ is None or (t_.dtype == torch.float64 and t_.is_contiguous())
        else:
            B = torch.empty((self.F, 6, 6), dtype=torch.float64, device=d) if want_cams else None
            gc = torch.empty((self.F, 6), dtype=torch.float64, device=d) if want_cams else None
            Cb = torch.empty((self.P, 6), dtype=torch.float64, device=d) if want_pts else None
            gp = torch.empty((self.P, 3), dtype=torch.float64, device=d) if want_pts else None
        self.ctx.check(lib.mm_ba_normal_eq(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(B), ptr(gc), ptr(Cb),
                                           ptr(gp)), "mm_ba_normal_eq")
        return B, gc, Cb, gp

    def jvp(self, cams, pts, wc, wp):
        out = torch.empty((self.O, 2), dtype=torch.float64, device=self.device)
        self.ctx.check(lib.mm_ba_jvp(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(wc), ptr(wp), ptr(out)),
                       "mm_ba_jvp")
        return out

    def jvp_dots(self, cams, pts, wc, wp, other=None):
        """jvp with its inner products fused in (mm_ba_jvp_dots): -> (out [O,2], rows [2,3]) with rows[0] = <out, other>
        (other None: <out, out>) and rows[1] = <out, out>, each as {0, total, total}."""
        out = torch.empty((self.O, 2), dtype=torch.float64, device=self.device)
        rows = torch.empty((2, 3), dtype=torch.float64, device=self.device)
        if getattr(self, "_jvp_ws", None) is None:
            self._jvp_ws = torch.zeros(lib.mm_ba_jvp_dots_workspace_bytes(C.byref(self.pb)), dtype=torch.uint8,
                                       device=self.device)
        self.ctx.check(lib.mm_ba_jvp_dots(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(wc), ptr(wp), ptr(out),
                                          ptr(other), ptr(rows), ptr(self._jvp_ws), self._jvp_ws.numel()), "mm_ba_jvp_dots")
        return out, rows

    def schur(self, cams, pts, Bd, Cd, gc, gp):
        """Reduced camera system.  With a pair list (banded problems) only the LOWER block band of S is produced
        (deterministically) and the rest of S is zero; otherwise all of S is filled."""
        n = 6 * self.F
        if self._S is None:
            self._alloc_S(n)
        S = self._S
        v = torch.empty(n, dtype=torch.float64, device=self.device)
        Cinv = torch.empty((self.P, 6), dtype=torch.float64, device=self.device)
        if self._schur_ws is None:
            nb = lib.mm_ba_schur_workspace_bytes(C.byref(self.pb))
            self._schur_ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=self.device)
        self.ctx.check(lib.mm_ba_schur(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(Bd), ptr(Cd), ptr(gc),
                                       ptr(gp), ptr(S), ptr(v), ptr(Cinv), ptr(self._schur_ws),
                                       self._schur_ws.numel()), "mm_ba_schur")
        return S, v, Cinv

    def _alloc_S(self, n):
        # n extra (zeroed) rows behind the matrix: band_view's diagonal-major strides run past row n - 1
        self._S_storage = torch.zeros(2 * n * n, dtype=torch.float64, device=self.device)
        self._S = self._S_storage[:n * n].view(n, n)

    def band_view(self, half_bandwidth):
        """The lower band of the reduced camera system as a strided [n, hb + 1] view of its storage: entry [j, k] is
        S[j + k, j] (rows past n - 1 fall into the zero padding).  What ranks exchange instead of the dense matrix."""
        n = 6 * self.F
        hb = int(min(half_bandwidth, n - 1))
        return torch.as_strided(self._S_storage, (n, hb + 1), (n + 1, n))

    def schur_solve(self, cams, pts, Bd, Cd, gc, gp, half_bandwidth):
        """Reduced camera system built and solved in one overlapped call (mm_ba_schur_solve): -> (info, dc [6F], Cinv).
        dc is the solution of S dc = v (the camera part of the damped Gauss-Newton step)."""
        n = 6 * self.F
        if self._S is None:
            self._alloc_S(n)
        v = torch.empty(n, dtype=torch.float64, device=self.device)
        Cinv = torch.empty((self.P, 6), dtype=torch.float64, device=self.device)
        info = torch.zeros(1, dtype=torch.int32, device=self.device)
        if self._schur_ws is None:
            nb = lib.mm_ba_schur_workspace_bytes(C.byref(self.pb))
            self._schur_ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=self.device)
        if self._chol_ws is None:
            self._chol_ws = torch.empty(lib.mm_chol_workspace_bytes(n), dtype=torch.uint8, device=self.device)
        hb = int(min(half_bandwidth, n))
        if self.slabs is not None and self.overlap:
            ns, cps, sseg, schunk = self.slabs
            a_seg, a_chunk = sseg.ctypes.data_as(_lib.c_i64p), schunk.ctypes.data_as(_lib.c_i64p)
        else:
            ns, cps, a_seg, a_chunk = 0, 0, None, None
        self.ctx.check(lib.mm_ba_schur_solve(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(Bd), ptr(Cd), ptr(gc),
                                             ptr(gp), ptr(self._S), ptr(v), ptr(Cinv), hb, ptr(info), ptr(self._schur_ws),
                                             self._schur_ws.numel(), ptr(self._chol_ws), self._chol_ws.numel(), ns, cps,
                                             a_seg, a_chunk), "mm_ba_schur_solve")
        return info, v, Cinv

    def multi_dot(self, pairs, split=0):
        """[k, 3] device tensor of inner products (camera part, point part, total); see ops.MultiDot."""
        if self._mdot is None:
            self._mdot = MultiDot(self.device, self.ctx)
        return self._mdot(pairs, split)

    _FUSED_ROWS = (2, 3, 2, 6, 0, 0)

    def trf_fused(self, op, ins, outs, scalars=(), h0=0.0, h1=0.0, split=0):
        """One fused element-wise pass of the trust-region step with its inner products (mm_trf_fused).
        -> device tensor [rows, 3] = (camera part, point part, total) per inner product, last row: maxima."""
        if self._mdot is None:
            self._mdot = MultiDot(self.device, self.ctx)
        ws = self._mdot.ws
        n = outs[0].numel()
        for t in list(ins) + list(outs) + list(scalars):
            assert t.dtype == torch.float64 and t.is_contiguous()
        pin = (C.c_void_p * len(ins))(*[t.data_ptr() for t in ins])
        pout = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        psc = (C.c_void_p * max(len(scalars), 1))(*[t.data_ptr() for t in scalars]) if scalars else None
        rows = self._FUSED_ROWS[op]
        if n == 0:
            return torch.zeros((max(rows, 1), 3), dtype=torch.float64, device=self.device)
        res = torch.empty((max(rows, 1), 3), dtype=torch.float64, device=self.device)
        self.ctx.check(lib.mm_trf_fused(self.ctx.h, op, pin, pout, psc, float(h0), float(h1), n, int(split), ptr(res), ptr(ws),
                                        ws.numel()), "mm_trf_fused")
        return res

    def trf_step2d(self, r0, d11, r1, r2, r3, bs, reg, info, Delta, board):
        """2-D trust-region subproblem on the device from the fused passes' results (mm_trf_step2d) -> board[0:14]."""
        self.ctx.check(lib.mm_trf_step2d(self.ctx.h, ptr(r0), ptr(d11), ptr(r1), ptr(r2), ptr(r3), ptr(bs), ptr(reg),
                                         ptr(info), float(Delta), ptr(board)), "mm_trf_step2d")

    def trf_damping(self, gh2, d11, Delta, min_damping):
        """Device scalars -> tensor [reg, max(reg, min_damping)] (see ops.trf_damping)."""
        return trf_damping(gh2, d11, Delta, min_damping, self.ctx)

    def chol_solve(self, S, v, half_bandwidth=None):
        """In-place banded Cholesky solve of the reduced camera system (see ops.chol_solve)."""
        return chol_solve(S, v, self.ctx, half_bandwidth=half_bandwidth)

    def scale_update(self, B, Cb, si, first):
        """si <- sqrt(diag(J^T J)) (first: zeros -> 1) or max(si, sqrt(diag)) in one launch (mm_ba_scale_update)."""
        self.ctx.check(lib.mm_ba_scale_update(self.ctx.h, self.F, self.P, ptr(B), ptr(Cb), ptr(si), 1 if first else 0),
                       "mm_ba_scale_update")
        return si

    def damp(self, B, Cb, si, reg, Bd, Cd):
        """Bd = B + reg diag(si_c^2), Cd = C + reg diag(si_p^2); reg: 1-element device tensor (mm_ba_damp)."""
        assert reg.dtype == torch.float64 and reg.numel() == 1
        self.ctx.check(lib.mm_ba_damp(self.ctx.h, self.F, self.P, ptr(B), ptr(Cb), ptr(si), ptr(reg), ptr(Bd), ptr(Cd)),
                       "mm_ba_damp")
        return Bd, Cd

    def chol_solve_sym(self, S, v, half_bandwidth, both_triangles):
        """Solution only (S destroyed): lets a narrow band be eliminated from both ends (see ops.chol_solve_sym)."""
        return chol_solve_sym(S, v, self.ctx, half_bandwidth, both_triangles)

    def trf_solve(self, cams, pts, ftol, xtol, gtol, max_nfev=None, min_damping=1e-9, log_cap=0):
        """The whole trust-region solve in one library call (mm_ba_trf; one GPU).  cams [F,6] / pts [P,3] are updated in
        place.  -> (report, rows) with rows = the (iteration, nfev, cost, reduction, step_norm, optimality) lines of
        SciPy's verbose=2 table (the first `log_cap` of them)."""
        for t in (cams, pts):
            assert t.dtype == torch.float64 and t.is_contiguous() and t.device == self.device
        if getattr(self, "_trf_ws", None) is None:
            self._trf_ws = torch.empty(lib.mm_ba_trf_workspace_bytes(C.byref(self.pb)), dtype=torch.uint8,
                                       device=self.device)
        prm = _lib.TrfParams(ftol, xtol, gtol, min_damping, int(max_nfev) if max_nfev else 0)
        rep = _lib.TrfReport()
        log = (_lib.TrfRow * max(log_cap, 1))()
        rc = lib.mm_ba_trf(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), C.byref(prm), C.byref(rep), log,
                           int(log_cap), ptr(self._trf_ws), self._trf_ws.numel())
        if rc and rep.status == -2:
            raise ValueError("Residuals are not finite in the initial point.")      # (as scipy's least_squares does)
        self.ctx.check(rc, "mm_ba_trf")
        rows = [(r.iteration, r.nfev, r.cost, r.reduction, r.step_norm, r.optimality)
                for r in log[:min(rep.log_rows, log_cap)]]
        return rep, rows

    def trf_solve_dist(self, cams, pts, ftol, xtol, gtol, allreduce, half_bandwidth, band_exchange, max_nfev=None,
                       min_damping=1e-9, log_cap=0):
        """mm_ba_trf_dist: the same loop sharded over ranks (this problem = the rank's points).  `allreduce` is a callable
        summing a device tensor over the ranks in place (parallel.AllReduce); the library calls back with pointers into its
        workspace, which are wrapped as tensor views here.  -> (report, rows) like trf_solve.
        An exception raised by `allreduce` -- or any error on this rank -- is FATAL FOR THE WHOLE GROUP: this rank leaves
        the loop while its peers wait inside their next collective (include/meatmodeler.h); it is re-raised here and the
        caller is expected to tear the process group down (its timeout bounds the peers' wait otherwise)."""
        for t in (cams, pts):
            assert t.dtype == torch.float64 and t.is_contiguous() and t.device == self.device
        need = lib.mm_ba_trf_dist_workspace_bytes(C.byref(self.pb), int(half_bandwidth))
        if getattr(self, "_trf_ws_dist", None) is None or self._trf_ws_dist.numel() < need:
            self._trf_ws_dist = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._trf_ws_dist
        base = ws.data_ptr()
        err = []

        def _cb(user, buf, count):
            try:
                off = int(buf) - base
                with torch.cuda.stream(self.ctx.stream):      # ordered on the stream the library launches on
                    allreduce(ws[off:off + 8 * int(count)].view(torch.float64))
                return 0
            except BaseException as e:      # (nothing may propagate through the C frames)
                err.append(e)
                return 1
        cb = _lib.ALLREDUCE_FN(_cb)
        d = _lib.Dist(int(getattr(allreduce, "rank", 0)), int(getattr(allreduce, "world_size", 1)), int(half_bandwidth),
                      1 if band_exchange else 0, cb, None)
        prm = _lib.TrfParams(ftol, xtol, gtol, min_damping, int(max_nfev) if max_nfev else 0)
        rep = _lib.TrfReport()
        log = (_lib.TrfRow * max(log_cap, 1))()
        rc = lib.mm_ba_trf_dist(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), C.byref(prm), C.byref(rep), log,
                                int(log_cap), ptr(ws), ws.numel(), C.byref(d))
        if err:
            raise err[0]
        if rc and rep.status == -2:
            raise ValueError("Residuals are not finite in the initial point.")
        self.ctx.check(rc, "mm_ba_trf_dist")
        rows = [(r.iteration, r.nfev, r.cost, r.reduction, r.step_norm, r.optimality)
                for r in log[:min(rep.log_rows, log_cap)]]
        return rep, rows

    def backsub(self, cams, pts, Cinv, gp, dc):
        dp = torch.empty((self.P, 3), dtype=torch.float64, device=self.device)
        if getattr(self, "_backsub_ws", None) is None:
            self._backsub_ws = torch.empty(lib.mm_ba_backsub_workspace_bytes(C.byref(self.pb)), dtype=torch.uint8,
                                           device=self.device)
        self.ctx.check(lib.mm_ba_backsub(self.ctx.h, C.byref(self.pb), ptr(cams), ptr(pts), ptr(Cinv), ptr(gp), ptr(dc),
                                         ptr(dp), ptr(self._backsub_ws), self._backsub_ws.numel()), "mm_ba_backsub")
        return dp


def trf_solve_batched(problems, cams_list, pts_list, ftol, xtol, gtol, max_nfev=None, min_damping=1e-9, ctx=None):
    """mm_ba_trf_batched: several independent problems (BADevice objects on one device) advanced in lock-step -- every
    kernel of the trust-region loop once per round for all of them.  cams_list[p] [F_p,6] / pts_list[p] [P_p,3] are
    updated in place; results are bit-identical to BADevice.trf_solve problem by problem.
    -> (list of reports, list of bool "was solved alone after the batch")."""
    n = len(problems)
    if n == 0:
        return [], []
    ctx = ctx or problems[0].ctx
    dev = problems[0].device
    pbs = (C.POINTER(_lib.BAProblem) * n)()
    cams_p, pts_p, ws_p = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    ws_b = (C.c_size_t * n)()
    keep = []
    for k, (pb, cm, pt) in enumerate(zip(problems, cams_list, pts_list)):
        for t in (cm, pt):
            assert t.dtype == torch.float64 and t.is_contiguous() and t.device == dev
        assert pb.device == dev
        need = lib.mm_ba_trf_batched_workspace_bytes(C.byref(pb.pb))
        if getattr(pb, "_trf_ws", None) is None or pb._trf_ws.numel() < need:
            pb._trf_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        pbs[k] = C.pointer(pb.pb)
        cams_p[k], pts_p[k], ws_p[k], ws_b[k] = cm.data_ptr(), pt.data_ptr(), pb._trf_ws.data_ptr(), pb._trf_ws.numel()
        keep.append(pb._trf_ws)
    prm = _lib.TrfParams(ftol, xtol, gtol, min_damping, int(max_nfev) if max_nfev else 0)
    reps = (_lib.TrfReport * n)()
    alone = (C.c_int32 * n)()
    rc = lib.mm_ba_trf_batched(ctx.h, n, pbs, cams_p, pts_p, C.byref(prm), reps, ws_p, ws_b, alone)
    if rc and any(r.status == -2 for r in reps):
        raise ValueError("Residuals are not finite in the initial point.")
    ctx.check(rc, "mm_ba_trf_batched")
    return list(reps), [bool(a) for a in alone]


class MultiDot:
    """[<a, b> for (a, b) in pairs] in one launch (mm_multi_dot): returns a device tensor [k, 3] =
    (sum over i < split, sum over i >= split, total).  Holds the zero-initialised workspace."""

    def __init__(self, device, ctx=None):
        self.ctx = ctx or default_context()
        self.ws = torch.zeros(lib.mm_multi_dot_workspace_bytes(), dtype=torch.uint8, device=device)

    def __call__(self, pairs, split=0):
        out_all = []
        for i0 in range(0, len(pairs), 8):
            chunk = pairs[i0:i0 + 8]
            k = len(chunk)
            n = chunk[0][0].numel()
            if n == 0:      # (empty tensors have no storage to point at)
                out_all.append(torch.zeros((k, 3), dtype=torch.float64, device=self.ws.device))
                continue
            for a, b in chunk:
                assert a.dtype == torch.float64 and b.dtype == torch.float64 and a.is_contiguous() and b.is_contiguous()
                assert a.numel() == n and b.numel() == n
            pa = (C.c_void_p * k)(*[a.data_ptr() for a, _ in chunk])
            pb_ = (C.c_void_p * k)(*[b.data_ptr() for _, b in chunk])
            out = torch.empty((k, 3), dtype=torch.float64, device=self.ws.device)
            self.ctx.check(lib.mm_multi_dot(self.ctx.h, k, pa, pb_, n, int(split), ptr(out), ptr(self.ws), self.ws.numel()),
                           "mm_multi_dot")
            out_all.append(out)
        return out_all[0] if len(out_all) == 1 else torch.cat(out_all)


def trf_damping(gh2, d11, Delta, min_damping, ctx=None):
    """Device scalars |g_h|^2 and |J_h g_h|^2 -> tensor [reg, max(reg, min_damping)] (mm_trf_damping)."""
    ctx = ctx or default_context()
    out = torch.empty(2, dtype=torch.float64, device=gh2.device)
    ctx.check(lib.mm_trf_damping(ctx.h, ptr(gh2), ptr(d11), float(Delta), float(min_damping), ptr(out)), "mm_trf_damping")
    return out


def chol_solve(A, b, ctx=None, half_bandwidth=None):
    """In place: A [n,n] f64 SPD (lower triangle used, overwritten by L), b [n] or [nrhs,n] overwritten by x.
    half_bandwidth: A[i][j] == 0 for i - j > half_bandwidth (None = dense).  Returns the device int32 info (0 = ok)."""
    ctx = ctx or default_context()
    n = A.shape[0]
    assert A.dtype == torch.float64 and A.is_contiguous() and b.is_contiguous() and b.shape[-1] == n
    nrhs = 1 if b.dim() == 1 else b.shape[0]
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    wsb = lib.mm_chol_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=A.device)
    hb = n if half_bandwidth is None else int(min(half_bandwidth, n))
    ctx.check(lib.mm_chol_solve(ctx.h, ptr(A), n, ptr(b), nrhs, hb, ptr(info), ptr(ws), wsb), "mm_chol_solve")
    return info


def chol_solve_sym(A, b, ctx=None, half_bandwidth=None, both_triangles=True):
    """In place, solution only: A [n,n] f64 SPD is destroyed, b [n] overwritten by x (mm_chol_solve_sym).
    both_triangles: both triangles of the band hold A (else only the lower one).  Returns the device int32 info."""
    ctx = ctx or default_context()
    n = A.shape[0]
    assert A.dtype == torch.float64 and A.is_contiguous() and b.is_contiguous() and b.dim() == 1 and b.shape[0] == n
    info = torch.zeros(1, dtype=torch.int32, device=A.device)
    wsb = lib.mm_chol_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=A.device)
    hb = n if half_bandwidth is None else int(min(half_bandwidth, n))
    ctx.check(lib.mm_chol_solve_sym(ctx.h, ptr(A), n, ptr(b), hb, 1 if both_triangles else 0, ptr(info), ptr(ws), wsb),
              "mm_chol_solve_sym")
    return info


# ------------------------------------------------------------------------------------------------ keyframe gating front end

def pyramid(img, max_level, ctx=None):
    """[H,W] u8 device tensor -> list of max_level + 1 device tensors (mm_pyr_down; level 0 is `img` itself)."""
    ctx = ctx or default_context()
    assert img.dtype == torch.uint8 and img.dim() == 2 and img.is_contiguous()
    levels = [img]
    for _ in range(int(max_level)):
        s = levels[-1]
        h, w = s.shape
        d = torch.empty(((h + 1) // 2, (w + 1) // 2), dtype=torch.uint8, device=s.device)
        ctx.check(lib.mm_pyr_down(ctx.h, ptr(s), w, h, s.stride(0), ptr(d), d.stride(0)), "mm_pyr_down")
        levels.append(d)
    return levels


def lk_track(prev_levels, next_levels, pts, win=(21, 21), max_count=30, epsilon=0.01, ctx=None):
    """Pyramidal Lucas-Kanade (mm_lk_track) on two pyramids from `pyramid`.  pts [n,2] f32 device.
    -> next [n,2] f32, status [n] u8, err [n] f32 (device)."""
    ctx = ctx or default_context()
    L = len(prev_levels)
    assert len(next_levels) == L and pts.dtype == torch.float32 and pts.is_contiguous() and pts.shape[-1] == 2
    n = pts.shape[0]
    d = pts.device
    out = torch.empty((n, 2), dtype=torch.float32, device=d)
    st = torch.empty(n, dtype=torch.uint8, device=d)
    err = torch.empty(n, dtype=torch.float32, device=d)
    if n == 0:
        return out, st, err
    arr_p = (C.c_void_p * L)(*[t.data_ptr() for t in prev_levels])
    arr_n = (C.c_void_p * L)(*[t.data_ptr() for t in next_levels])
    ws = (C.c_int * L)(*[t.shape[1] for t in prev_levels])
    hs = (C.c_int * L)(*[t.shape[0] for t in prev_levels])
    ps = (C.c_int * L)(*[t.stride(0) for t in prev_levels])
    for a, b in zip(prev_levels, next_levels):
        assert a.shape == b.shape and a.stride(0) == b.stride(0)
    eps = min(max(float(epsilon), 0.0), 10.0) ** 2
    ctx.check(lib.mm_lk_track(ctx.h, arr_p, arr_n, ws, hs, ps, L, ptr(pts), n, int(win[0]), int(win[1]),
                              min(max(int(max_count), 0), 100), eps, ptr(out), ptr(st), ptr(err)), "mm_lk_track")
    return out, st, err


def min_eig(img, block_size=3, ctx=None):
    """Shi-Tomasi minimum-eigenvalue map [H,W] f64 (mm_min_eig)."""
    ctx = ctx or default_context()
    assert img.dtype == torch.uint8 and img.dim() == 2 and img.stride(1) == 1
    h, w = img.shape
    eig = torch.empty((h, w), dtype=torch.float64, device=img.device)
    ctx.check(lib.mm_min_eig(ctx.h, ptr(img), w, h, img.stride(0), int(block_size), ptr(eig)), "mm_min_eig")
    return eig


def good_features(img, max_corners, quality, min_distance, block_size=3, ctx=None):
    """cv2.goodFeaturesToTrack on a device image: eigenvalue map, threshold + non-maximum suppression and compaction on
    the device, ordering by torch.sort (index plumbing), greedy minimum-distance selection on the host
    (mm_gftt_select).  -> numpy [n,2] f32 (x, y)."""
    ctx = ctx or default_context()
    h, w = img.shape
    if h < 3 or w < 3:                            # no interior pixel: no corner (the 3x3 suppression needs a neighbourhood)
        return np.zeros((0, 2), np.float32)
    eig = min_eig(img, block_size, ctx)
    cap = h * w                                  # (a plateau of equal values keeps every one of its pixels)
    d = img.device
    mx = torch.zeros(1, dtype=torch.int64, device=d)
    vb = torch.empty(cap, dtype=torch.int64, device=d)
    pos = torch.empty(cap, dtype=torch.int32, device=d)
    cnt = torch.zeros(1, dtype=torch.int32, device=d)
    ctx.check(lib.mm_corner_candidates(ctx.h, ptr(eig), w, h, float(quality), ptr(mx), ptr(vb), ptr(pos), cap, ptr(cnt)),
              "mm_corner_candidates")
    n = min(int(cnt.item()), cap)
    if n == 0:
        return np.zeros((0, 2), np.float32)
    pos_s, o1 = torch.sort(pos[:n], stable=True)                       # by position ...
    _, o2 = torch.sort(vb[:n][o1], descending=True, stable=True)       # ... then stably by strength (descending)
    pos_h = np.ascontiguousarray(pos_s[o2].cpu().numpy(), np.int32)
    out_cap = int(max_corners) if max_corners and max_corners > 0 else n
    out = np.zeros((max(out_cap, 1), 2), np.float32)
    m = lib.mm_gftt_select(pos_h.ctypes.data_as(C.c_void_p), n, w, h, int(max_corners or 0), float(min_distance),
                           out.ctypes.data_as(C.c_void_p), out_cap)
    if m < 0:
        raise _lib.MMError(f"mm_gftt_select failed ({m})")
    return out[:m]


_LAB_TABLES = {}


def increase_contrast(bgr, clip_limit=3.5, tiles=(8, 8), want_grey=False, ctx=None):
    """bgr [B,H,W,3] u8 device -> contrast-enhanced BGR (and the grey image of the result): mm_increase_contrast."""
    ctx = ctx or default_context()
    assert bgr.dtype == torch.uint8 and bgr.dim() == 4 and bgr.shape[-1] == 3 and bgr.is_contiguous()
    B, H, W, _ = bgr.shape
    d = bgr.device
    key = str(d)
    if key not in _LAB_TABLES:
        from .frame_tables import lab_tables
        _LAB_TABLES[key] = tuple(torch.as_tensor(t.view(np.int16) if t.dtype == np.uint16 else t).to(d) for t in lab_tables())
    g, cb, gi = _LAB_TABLES[key]
    out = torch.empty_like(bgr)
    grey = torch.empty((B, H, W), dtype=torch.uint8, device=d) if want_grey else None
    wsb = lib.mm_contrast_workspace_bytes(B, W, H, int(tiles[0]), int(tiles[1]))
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=d)
    ctx.check(lib.mm_increase_contrast(ctx.h, ptr(bgr), B, W, H, ptr(g), ptr(cb), ptr(gi), float(clip_limit), int(tiles[0]),
                                       int(tiles[1]), ptr(out), ptr(grey), ptr(ws), ws.numel()), "mm_increase_contrast")
    return (out, grey) if want_grey else out


def bgr_to_grey(bgr, ctx=None):
    """[...,3] u8 device -> [...] u8: cv2.COLOR_BGR2GRAY's fixed-point weights (mm_bgr_to_grey)."""
    ctx = ctx or default_context()
    assert bgr.dtype == torch.uint8 and bgr.shape[-1] == 3 and bgr.is_contiguous()
    grey = torch.empty(bgr.shape[:-1], dtype=torch.uint8, device=bgr.device)
    ctx.check(lib.mm_bgr_to_grey(ctx.h, ptr(bgr), grey.numel(), ptr(grey)), "mm_bgr_to_grey")
    return grey
