"""Batched clip pipeline: the hot path of `processor.process` (reference processor.py:356-470) run over a whole
clip with device-resident state — detect all frames, match all consecutive pairs, link tracks, triangulate, bundle
adjust.  Same stages and semantics as the per-keyframe drop-in functions in `processor.py`, but frames, descriptors
and key points stay in HBM and each stage is a few large launches instead of one small launch per keyframe.

Keyframe gating, calibration and PnP are outside the scope: the caller provides K and one extrinsic
per frame (the reference gets them from calibrate / poseEstimation / adjustPose, processor.py:422-448).
"""
import ctypes as C
import queue
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib, ops, parallel
from ._lib import lib, default_context, c_i32p, c_i64p, c_f32p, MMError
from .bundleAdjuster import SchurTRF, frameParameters
from .orb_pattern import brief_pattern


class ClipPipeline:
    def __init__(self, height, width, nfeatures, batch=32, ratio=0.75, device=None, ctx=None, nlevels=8):
        self.ctx = ctx or default_context()
        self.device = device or self.ctx.device
        self.H, self.W = height, width
        self.prm = ops.orb_params(nfeatures, nlevels=nlevels)
        self.cap = nfeatures
        self.batch = batch
        self.ratio = ratio
        self.wsp = ops.OrbWorkspace(batch, height, width, self.prm, self.device, brief_pattern())
        self.scales = ops.orb_level_sizes(height, width, self.prm)[3]

    # ------------------------------------------------------------------------------------------- detect
    def detect(self, frames):
        """frames [F,H,W] u8 on the device -> dict of per-frame tables (xy [F,cap,2] f32, desc [F,cap,32] u8, n [F],
        meta, resp, mom), all device tensors."""
        F = frames.shape[0]
        d = self.device
        out = dict(xy=torch.zeros((F, self.cap, 2), dtype=torch.float32, device=d),
                   meta=torch.zeros((F, self.cap, 4), dtype=torch.int32, device=d),
                   resp=torch.zeros((F, self.cap), dtype=torch.float32, device=d),
                   mom=torch.zeros((F, self.cap, 2), dtype=torch.int32, device=d),
                   desc=torch.zeros((F, self.cap, 32), dtype=torch.uint8, device=d),
                   n=torch.zeros(F, dtype=torch.int32, device=d))
        for b0 in range(0, F, self.batch):
            b1 = min(F, b0 + self.batch)
            ops.orb_detect_compute(frames[b0:b1], self.wsp, self.ctx,
                                   out=(out["xy"][b0:b1], out["meta"][b0:b1], out["resp"][b0:b1], out["mom"][b0:b1],
                                        out["desc"][b0:b1], out["n"][b0:b1]))
        return out

    # ------------------------------------------------------------------------------------------- match
    def match(self, det, pair_lo=0, pair_hi=None):
        """Consecutive pairs (k, k+1), k in [pair_lo, pair_hi) relative to the frames in `det`:
        query = frame k, train = frame k+1 (processor.py:133).  -> pairs [n_pairs, cap, 2] i32, m [n_pairs] i32."""
        F = det["desc"].shape[0]
        if pair_hi is None:
            pair_hi = F - 1
        npairs = max(pair_hi - pair_lo, 0)
        desc, n = det["desc"], det["n"]
        if npairs == 0:
            return (torch.zeros((0, self.cap, 2), dtype=torch.int32, device=self.device),
                    torch.zeros(0, dtype=torch.int32, device=self.device))
        q = desc[pair_lo:pair_hi]
        t = desc[pair_lo + 1:pair_hi + 1]
        idx, dist = ops.bf_knn2_batched(q, t, n[pair_lo:pair_hi].contiguous(), n[pair_lo + 1:pair_hi + 1].contiguous(),
                                        self.ctx)
        return ops.ratio_filter_batched(idx, dist, self.ratio, n[pair_lo:pair_hi].contiguous(), self.ctx)

    # ------------------------------------------------------------------------------------------- link
    def link(self, kp_count, kp_xy, match_count, matches):
        """Host track linking over the clip (mm_link_tracks_clip).  numpy inputs:
        kp_count [F] i32, kp_xy [F,cap,2] f32, match_count [F-1] i32, matches [F-1,mcap,2] i32.
        -> (track_ptr [T+1] i64, obs_frame [O] i32, obs_kp [O] i32)."""
        kp_count = np.ascontiguousarray(kp_count, np.int32)
        kp_xy = np.ascontiguousarray(kp_xy, np.float32)
        match_count = np.ascontiguousarray(match_count, np.int32)
        matches = np.ascontiguousarray(matches, np.int32)
        F = len(kp_count)
        cap = kp_xy.shape[1] if kp_xy.ndim == 3 else 0
        mcap = matches.shape[1] if matches.ndim == 3 else 0
        total = int(match_count.sum())
        track_ptr = np.zeros(total + 2, np.int64)
        obs_frame = np.zeros(2 * total + 2, np.int32)
        obs_kp = np.zeros(2 * total + 2, np.int32)
        n_obs = C.c_int64(0)
        nt = lib.mm_link_tracks_clip(F, cap, kp_count.ctypes.data_as(c_i32p), kp_xy.ctypes.data_as(c_f32p), mcap,
                                     match_count.ctypes.data_as(c_i32p), matches.ctypes.data_as(c_i32p), total + 1,
                                     2 * total + 1, track_ptr.ctypes.data_as(c_i64p), obs_frame.ctypes.data_as(c_i32p),
                                     obs_kp.ctypes.data_as(c_i32p), C.byref(n_obs))
        if nt < 0:
            raise MMError(f"mm_link_tracks_clip failed ({nt})")
        return track_ptr[:nt + 1], obs_frame[:n_obs.value], obs_kp[:n_obs.value]

    # ------------------------------------------------------------------------------------------- triangulate
    def triangulate(self, track_ptr, obs_frame, obs_kp, kp_xy_dev, projections):
        """First/last observation of every track -> 3-D points [T,3] f64 on the device (processor.py:246-261).
        track_ptr / obs_frame / obs_kp: device tensors (or numpy arrays, uploaded here)."""
        d = self.device
        track_ptr, obs_frame, obs_kp = (t if torch.is_tensor(t) else torch.as_tensor(np.ascontiguousarray(t)).to(d)
                                        for t in (track_ptr, obs_frame, obs_kp))
        T = track_ptr.shape[0] - 1
        if T <= 0:
            return torch.zeros((0, 3), dtype=torch.float64, device=d)
        first = track_ptr[:-1].long()
        last = track_ptr[1:].long() - 1
        f0, f1 = obs_frame[first].int(), obs_frame[last].int()
        x0 = kp_xy_dev[f0.long(), obs_kp[first].long()].to(torch.float64)
        x1 = kp_xy_dev[f1.long(), obs_kp[last].long()].to(torch.float64)
        proj = torch.as_tensor(np.ascontiguousarray(projections, np.float64)).to(d)
        return ops.triangulate_dlt(proj, f0.contiguous(), f1.contiguous(), x0, x1, self.ctx)

    # ------------------------------------------------------------------------------------------- flatten (managePoints)
    @staticmethod
    def flatten(track_ptr, obs_frame, obs_kp, kp_xy_host):
        """(coordinates [O,2] f64, frame_indices [O] i32, point_indices [O] i32) in managePoints order
        (processor.py:264-291): point-major, insertion order inside a track."""
        lens = np.diff(track_ptr).astype(np.int64)
        pi = np.repeat(np.arange(len(lens), dtype=np.int32), lens)
        coords = kp_xy_host[obs_frame, obs_kp].astype(np.float64)
        return coords, obs_frame.astype(np.int32), pi

    # ------------------------------------------------------------------------------------------- sliding-window BA
    @staticmethod
    def window_selection(first_frame, last_frame, lo, hi, n_frames):
        """Tracks adjusted by the window of keyframes [lo, hi): every observation inside the window, and the track is
        finished ("popped", processor.py:233-238) by keyframe hi - 1 -- its last observation is at or before hi - 2 --
        or the clip ends with this window.  Works on numpy arrays and on tensors alike."""
        inside = (first_frame >= lo) & (last_frame < hi)
        if hi >= n_frames:
            return inside
        return inside & (last_frame <= hi - 2)

    def adjust_windows(self, out, K, extrinsics, window=50, stride=25, ftol=1e-4, verbose=0, timers=None, dist=None,
                       order="sequential", streams=1, batched=False, max_nfev=None):
        """Incremental bundle adjustment over a sliding window of keyframes: the reference keeps this step as a
        commented hook (processor.py:395-408: after a keyframe whose tracks were popped, `managePoints(popped_tracks)`
        + `adjustPoints` over everything so far); bounding it to the last `window` keyframes is what makes the
        2000-frame clips tractable.  For hi = window, window + stride, ..., F the cameras
        lo = hi - window .. hi - 1 and the finished tracks that live entirely inside the window are adjusted with
        exactly the solver of `adjustPoints` (all window cameras and points free, ftol as given); camera parameters and
        points are written back and later windows start from them.

        With `dist` (torch.distributed, every rank holding the same linked tracks as `run(..., dist=dist)` leaves them)
        each window's points are split over the ranks by observation count, cameras are replicated, and the solver
        all-reduces the camera-side blocks exactly as the global adjustment does ("C5 sliding
        window: same, with F replaced by W"); the adjusted points of a window are re-assembled on every rank.

        order = "wavefront": the windows are coloured so that windows of one colour share no camera (two colours for
        stride >= window / 2: the even windows, then the odd ones, which start from both neighbours' results) and each
        colour is one pass in which every window is solved WHOLE by one rank (round robin), the results of a pass being
        exchanged with one all-reduce of the touched cameras / points.  Windows of a pass are independent, so this
        scales with the number of GPUs instead of paying a collective per trust-region iteration of a 300-unknown
        system, and the result does not depend on the world size at all (bit for bit: every window is solved by the
        same un-sharded code on the same inputs).  It is a different -- equally valid -- schedule than "sequential".
        streams > 1 (wavefront only): that many windows of a pass are in flight on this GPU at once, one HIP stream and
        one host thread each; the result does not depend on it either.
        max_nfev: evaluation budget PER WINDOW (None: SciPy's default of 100 n, as adjustPoints).  A window is adjusted again
        by its successor, so a budget is a reasonable guard against the occasional outlier-laden window that crawls
        (observed: one window of 79 taking 3466 evaluations where the others take 30 - 340); a window that exhausts it
        reports status 0 and its last accepted iterate is used.
        batched = True (wavefront only; supersedes `streams`): ALL windows of a pass that this rank owns advance in
        lock-step through mm_ba_trf_batched -- every kernel of the trust-region loop is launched once per round for all of
        them, so a pass costs about as many evaluations as its slowest window needs instead of the sum over its windows.
        Bit-identical to solving the windows one at a time.

        `out` is the result of `run(..., ba=False)`.  -> dict(cams [F,6] device, points [T,3] device, windows=[...])."""
        d = self.device
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        allreduce = parallel.AllReduce() if world > 1 else None
        if order not in ("sequential", "wavefront"):
            raise ValueError("order must be 'sequential' or 'wavefront'")
        if order == "wavefront":
            return self._adjust_windows_wavefront(out, K, extrinsics, window, stride, ftol, verbose, timers, allreduce,
                                                  world, rank, streams, batched, max_nfev)
        F = int(np.asarray(extrinsics).shape[0])
        tp, of_, ok = out["track_ptr_dev"], out["obs_frame_dev"], out["obs_kp_dev"]
        xy = out["xy_dev"]
        T = tp.shape[0] - 1
        with np.errstate(all="ignore"):
            cams = torch.as_tensor(frameParameters(np.asarray(extrinsics, float)[:, :3, :]).reshape(F, 6)).to(d)
        pts = out["points0"].clone()
        stats = []
        if T == 0:
            return dict(cams=cams, points=pts, windows=stats)
        tp64 = tp.long()
        first_f, last_f = of_[tp64[:-1]], of_[tp64[1:] - 1]
        lens_all = tp64[1:] - tp64[:-1]
        window = max(2, min(int(window), F))
        his = list(range(window, F, max(1, int(stride)))) + [F]
        t0 = time.perf_counter()
        for hi in his:
            lo = max(0, hi - window)
            sel = torch.nonzero(self.window_selection(first_f, last_f, lo, hi, F)).reshape(-1)
            P = int(sel.numel())
            if P == 0:
                continue
            lens = lens_all[sel]
            P_all, p_lo, p_hi = P, 0, P
            if world > 1 and P >= 4 * world:
                # this rank's contiguous share of the window's points (balanced by observations); tiny windows are
                # solved redundantly on every rank (identical, deterministic) rather than with empty shards
                cum = torch.cat([torch.zeros(1, dtype=torch.int64, device=d), torch.cumsum(lens, 0)])
                bounds = parallel.split_by_weight(cum, world)
                p_lo, p_hi = bounds[rank], bounds[rank + 1]
                sel_all, sel = sel, sel[p_lo:p_hi]
                lens = lens[p_lo:p_hi]
                P = p_hi - p_lo
            sharded = world > 1 and P_all >= 4 * world
            # the selected tracks' observations, point-major (managePoints order; mm_flatten_tracks)
            coords, fi, pi = ops.flatten_tracks(tp, of_, ok, xy, sel=sel, frame_offset=lo, ctx=self.ctx)
            O = int(fi.numel())
            pb = ops.BADevice(K, fi, pi, coords, hi - lo, P, d, self.ctx)
            res = SchurTRF(pb, allreduce=allreduce if sharded else None).solve(
                cams[lo:hi].contiguous(), pts[sel].contiguous(), ftol=ftol, max_nfev=max_nfev, verbose=verbose if rank == 0 else 0)
            cams[lo:hi] = res.cams
            if sharded:
                buf = torch.zeros((P_all, 3), dtype=torch.float64, device=d)
                buf[p_lo:p_hi] = res.pts
                allreduce(buf)                                  # disjoint shards: the sum re-assembles the window
                pts[sel_all] = buf
                O_all = torch.tensor([float(O)], dtype=torch.float64, device=d)
                allreduce(O_all)
                O = int(O_all.item())
            else:
                pts[sel] = res.pts
            stats.append(dict(lo=lo, hi=hi, points=P_all, observations=O, nfev=res.nfev, cost=res.cost,
                              status=res.status))
        self.ctx.sync()
        if timers is not None:
            timers["ba_windows"] = timers.get("ba_windows", 0.0) + (time.perf_counter() - t0) * 1e3
        return dict(cams=cams, points=pts, windows=stats)

    def _window_problem(self, out, first_f, last_f, lens_all, tp64, lo, hi, F, ctx=None):
        """Selection + managePoints-order flattening of one window on the device -> (sel, fi, pi, coords, P, O)."""
        d = self.device
        of_, ok, xy = out["obs_frame_dev"], out["obs_kp_dev"], out["xy_dev"]
        sel = torch.nonzero(self.window_selection(first_f, last_f, lo, hi, F)).reshape(-1)
        P = int(sel.numel())
        if P == 0:
            return sel, None, None, None, 0, 0
        # (mm_flatten_offsets + mm_flatten_tracks: the window's tracks, frame indices relative to the window)
        coords, fi, pi = ops.flatten_tracks(out["track_ptr_dev"], of_, ok, xy, sel=sel, frame_offset=lo, ctx=ctx or self.ctx)
        return sel, fi, pi, coords, P, int(fi.numel())

    def _adjust_windows_wavefront(self, out, K, extrinsics, window, stride, ftol, verbose, timers, allreduce, world, rank,
                                  streams=1, batched=False, max_nfev=None):
        d = self.device
        F = int(np.asarray(extrinsics).shape[0])
        tp, of_ = out["track_ptr_dev"], out["obs_frame_dev"]
        T = tp.shape[0] - 1
        with np.errstate(all="ignore"):
            cams = torch.as_tensor(frameParameters(np.asarray(extrinsics, float)[:, :3, :]).reshape(F, 6)).to(d)
        pts = out["points0"].clone()
        if T == 0:
            return dict(cams=cams, points=pts, windows=[])
        tp64 = tp.long()
        first_f, last_f = of_[tp64[:-1]], of_[tp64[1:] - 1]
        lens_all = tp64[1:] - tp64[:-1]
        window = max(2, min(int(window), F))
        his = list(range(window, F, max(1, int(stride)))) + [F]
        wins = [(max(0, hi - window), hi) for hi in his]
        # greedy colouring in window order: a colour's windows share no camera
        colour_end, colours = [], []
        for lo, hi in wins:
            c = next((k for k, e in enumerate(colour_end) if e <= lo), len(colour_end))
            if c == len(colour_end):
                colour_end.append(hi)
            else:
                colour_end[c] = hi
            colours.append(c)
        t0 = time.perf_counter()
        table = torch.zeros((len(wins), 5), dtype=torch.float64, device=d)      # points, observations, nfev, status, cost
        # Windows of a pass are independent AND small (a 300-unknown reduced system keeps 16 workgroups busy): this rank
        # solves up to `streams` of them at once, each on its own HIP stream with its own library context, driven by its
        # own host thread (mm_ba_trf blocks in C with the GIL released).  Every window still runs the same code on the
        # same inputs, so the result does not depend on `streams`.
        n_streams = max(1, int(streams))
        slots = queue.SimpleQueue()
        if n_streams > 1:
            for _ in range(n_streams):
                st = torch.cuda.Stream(device=d)
                slots.put((st, _lib.Context(d, st)))

        def solve_window(k, ctx, cams, pts, cams_upd, cam_mask, pts_upd, pt_mask):
            lo, hi = wins[k]
            sel, fi, pi, coords, P, O = self._window_problem(out, first_f, last_f, lens_all, tp64, lo, hi, F, ctx)
            if P == 0:
                return
            pb = ops.BADevice(K, fi, pi, coords, hi - lo, P, d, ctx)
            res = SchurTRF(pb).solve(cams[lo:hi].contiguous(), pts[sel].contiguous(), ftol=ftol, max_nfev=max_nfev,
                                     verbose=verbose if rank == 0 and n_streams == 1 else 0)
            cams_upd[lo:hi] = res.cams
            cam_mask[lo:hi] = 1.0
            pts_upd[sel] = res.pts
            pt_mask[sel] = 1.0
            table[k] = torch.tensor([P, O, res.nfev, res.status, res.cost], dtype=torch.float64, device=d)

        def solve_window_on_slot(k, *state):
            slot = slots.get()
            try:
                with torch.cuda.stream(slot[0]):
                    solve_window(k, slot[1], *state)
                    slot[0].synchronize()       # results are in place before the pass's thread pool is joined
            finally:
                slots.put(slot)

        for c in range(len(colour_end)):
            mine = [k for j, k in enumerate(k for k, cc in enumerate(colours) if cc == c) if j % world == rank]
            cams_upd, pts_upd = torch.zeros_like(cams), torch.zeros_like(pts)
            cam_mask = torch.zeros(F, dtype=torch.float64, device=d)
            pt_mask = torch.zeros(T, dtype=torch.float64, device=d)
            state = (cams, pts, cams_upd, cam_mask, pts_upd, pt_mask)
            if batched and len(mine) > 1:
                # every window of the pass through ONE lock-step solve (mm_ba_trf_batched)
                items = []
                for k in mine:
                    lo, hi = wins[k]
                    sel, fi, pi, coords, P, O = self._window_problem(out, first_f, last_f, lens_all, tp64, lo, hi, F)
                    if P == 0:
                        continue
                    pb = ops.BADevice(K, fi, pi, coords, hi - lo, P, d, self.ctx)
                    items.append((k, sel, pb, cams[lo:hi].contiguous(), pts[sel].contiguous(), P, O))
                reps, _ = ops.trf_solve_batched([it[2] for it in items], [it[3] for it in items], [it[4] for it in items],
                                                ftol, 1e-8, 1e-8, max_nfev=max_nfev, ctx=self.ctx)
                rows, ks = [], []
                for (k, sel, pb, cw, pw, P, O), rep in zip(items, reps):
                    lo, hi = wins[k]
                    cams_upd[lo:hi] = cw
                    cam_mask[lo:hi] = 1.0
                    pts_upd[sel] = pw
                    pt_mask[sel] = 1.0
                    rows.append([P, O, rep.nfev, rep.status, rep.cost])
                    ks.append(k)
                if ks:
                    table[torch.tensor(ks, device=d)] = torch.tensor(rows, dtype=torch.float64, device=d)
            elif n_streams > 1 and len(mine) > 1:
                torch.cuda.current_stream(d).synchronize()        # the pass's inputs are complete
                with ThreadPoolExecutor(max_workers=n_streams) as pool:
                    for f_ in [pool.submit(solve_window_on_slot, k, *state) for k in mine]:
                        f_.result()
            else:
                for k in mine:
                    solve_window(k, self.ctx, *state)
            if allreduce is not None:          # the windows of a pass touch disjoint cameras / points: the sums are copies
                for t_ in (cams_upd, cam_mask, pts_upd, pt_mask):
                    allreduce(t_)
            cams = torch.where(cam_mask[:, None] > 0, cams_upd, cams)
            pts = torch.where(pt_mask[:, None] > 0, pts_upd, pts)
        if allreduce is not None:
            allreduce(table)
        self.ctx.sync()
        if timers is not None:
            timers["ba_windows"] = timers.get("ba_windows", 0.0) + (time.perf_counter() - t0) * 1e3
        tab = table.cpu().numpy()
        stats = [dict(lo=wins[k][0], hi=wins[k][1], points=int(r[0]), observations=int(r[1]), nfev=int(r[2]), status=int(r[3]),
                      cost=float(r[4]), colour=colours[k]) for k, r in enumerate(tab) if r[0] > 0]
        return dict(cams=cams, points=pts, windows=stats)

    @staticmethod
    def tracks_to_host(out):
        """Copy the linked tracks of a `run` result to the host: adds numpy `track_ptr` [T+1] i64, `obs_frame`,
        `obs_kp` [O] i32 (the layout mm_link_tracks_clip returns) to `out`."""
        out["track_ptr"] = out["track_ptr_dev"].cpu().numpy().astype(np.int64)
        out["obs_frame"] = out["obs_frame_dev"].cpu().numpy()
        out["obs_kp"] = out["obs_kp_dev"].cpu().numpy()
        return out

    # ------------------------------------------------------------------------------------------- whole clip
    def run(self, frames, K, extrinsics, ba=True, ftol=1e-4, verbose=0, dist=None, timers=None, max_nfev=None,
            force_collectives=False):
        """frames [F,H,W] u8 (device).  With `dist` = torch.distributed (initialised), frames are the FULL clip on
        every rank (synthetic input is generated locally) and the work is sharded as described in parallel.py.
        `max_nfev`: evaluation budget of the adjustment (None = SciPy's default, as adjustPoints).
        `force_collectives`: take the gather / all-reduce path even in a one-rank group (the collectives are trivial
        but travel the real backend: how backend "nccl" is exercised on a single GPU)."""
        F = frames.shape[0]
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        sharded = world > 1 or (dist is not None and force_collectives)
        T = timers if timers is not None else {}

        def tic(name):
            self.ctx.sync()
            T.setdefault("_open", {})[name] = time.perf_counter()

        def toc(name):
            self.ctx.sync()
            T[name] = T.get(name, 0.0) + (time.perf_counter() - T["_open"].pop(name)) * 1e3

        (p_lo, p_hi), (f_lo, f_hi) = parallel.pair_block(F, rank, world)
        tic("detect")
        det = self.detect(frames[f_lo:f_hi]) if f_hi > f_lo else None
        toc("detect")
        tic("match")
        if det is not None:
            pairs, m = self.match(det)
        else:
            pairs = torch.zeros((0, self.cap, 2), dtype=torch.int32, device=self.device)
            m = torch.zeros(0, dtype=torch.int32, device=self.device)
        toc("match")
        tic("link")
        if sharded:
            # every rank needs every frame's key points and every pair's matches to link identical tracks.  The block
            # partition is arithmetic on (F, world), so every rank knows every rank's row counts: fixed-shape device
            # all-gathers (RCCL over xGMI under "nccl"), nothing goes through host memory.
            blocks = [parallel.pair_block(F, r, world) for r in range(world)]
            pair_counts = [b[0][1] - b[0][0] for b in blocks]
            last_rank_with_pairs = max(r for r in range(world) if pair_counts[r] > 0)
            # frames f_lo .. f_lo+own-1 are owned; the halo frame belongs to the next rank (the last one keeps it)
            frame_counts = [pair_counts[r] + (1 if r == last_rank_with_pairs else 0) for r in range(world)]
            keep = frame_counts[rank]
            d = self.device
            if det is not None:
                xy_l, n_l = det["xy"][:keep], det["n"][:keep]
            else:
                xy_l = torch.zeros((0, self.cap, 2), dtype=torch.float32, device=d)
                n_l = torch.zeros(0, dtype=torch.int32, device=d)
            xy_dev = parallel.gather_blocks(xy_l.contiguous(), frame_counts, dist)
            n_d = parallel.gather_blocks(n_l.contiguous(), frame_counts, dist)
            m_d = parallel.gather_blocks(m.contiguous(), pair_counts, dist)
            pairs_d = parallel.gather_blocks(pairs.contiguous(), pair_counts, dist)
        else:
            xy_dev, n_d, m_d, pairs_d = det["xy"], det["n"], m, pairs
        m_h = m_d.cpu().numpy()
        n_h = n_d.cpu().numpy()
        track_ptr, obs_frame, obs_kp, bad = ops.link_tracks_device(n_d, xy_dev, m_d, pairs_d, self.ctx)
        if bad:
            raise MMError("link: match index outside the key point tables")
        toc("link")
        ext = np.asarray(extrinsics, float)[:, :3, :]
        proj = np.einsum("ij,fjk->fik", np.asarray(K, float), ext)   # K [R|t], processor.py:184,448
        tic("triangulate")
        X = self.triangulate(track_ptr, obs_frame, obs_kp, xy_dev, proj)
        toc("triangulate")
        P, O = track_ptr.shape[0] - 1, obs_frame.shape[0]
        out = dict(det=det, n_tracks=P, n_obs=O, points0=X, track_ptr_dev=track_ptr, obs_frame_dev=obs_frame,
                   obs_kp_dev=obs_kp, xy_dev=xy_dev, match_count=m_h, kp_count=n_h, pairs_local=int(p_hi - p_lo),
                   frames_local=int(f_hi - f_lo))
        if not ba or O == 0:
            T.pop("_open", None)
            return out
        tic("ba")
        with np.errstate(all="ignore"):
            cams0 = frameParameters(ext).reshape(F, 6)
        # managePoints order (point-major, insertion order inside a track), assembled on the device
        d = self.device
        if world > 1:
            lo, hi, o_lo, o_hi = parallel.partition_tracks(track_ptr, rank, world)
        else:
            lo, hi, o_lo, o_hi = 0, P, 0, O
        # (mm_flatten_tracks: this rank's tracks lo .. hi - 1)
        coords_d, of_d, pi_d = ops.flatten_tracks(track_ptr, obs_frame, obs_kp, xy_dev, t_lo=lo, n_sel=hi - lo, ctx=self.ctx)
        pb = ops.BADevice(K, of_d, pi_d, coords_d, F, hi - lo, d, self.ctx)
        pts0 = X[lo:hi].contiguous()
        solver = SchurTRF(pb, allreduce=parallel.AllReduce(force=force_collectives) if sharded else None)
        cams_d = torch.as_tensor(cams0).to(self.device)
        tic("ba_solve")
        res = solver.solve(cams_d, pts0, ftol=ftol, max_nfev=max_nfev, verbose=verbose if rank == 0 else 0)
        toc("ba_solve")
        toc("ba")
        out.update(ba=res, n_obs_local=pb.O, cam_span=pb.cam_span, n_pairs=pb.n_pairs)
        T.pop("_open", None)
        return out
