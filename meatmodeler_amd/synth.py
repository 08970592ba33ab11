"""Seeded synthetic inputs for the hot path.

Nothing here is taken from the reference: the reference has no sample data and no tests
(SURVEY.md §4).  Everything is generated from ``numpy.random.default_rng(seed)`` so the CPU
oracle and the HIP path see byte-identical inputs.

* ``orbit_cameras``      – pinhole K and world->camera extrinsics on an orbit arc.
* ``make_ba_problem``    – tracks of L consecutive frames over an orbit (BA micro workload).
* ``random_descriptors`` – planted-match descriptor pairs for the matcher (C2 recipe).
* ``make_texture`` / ``render_orbit_frames`` – procedural textured scene rendered to u8 grey frames.
"""
import numpy as np


def default_K(width=1920, height=1080, f=None):
    if f is None:
        f = 1500.0 * width / 1920.0
    return np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])


def look_at_extrinsic(C, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """World->camera [R|t] (3x4) of a camera at C looking at target (x right, y down, z forward)."""
    C = np.asarray(C, float)
    z = np.asarray(target, float) - C
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, float))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    t = -R @ C
    return np.hstack([R, t[:, None]])


def orbit_cameras(n_frames, arc_deg=360.0, radius=10.0, height=2.0):
    """n_frames extrinsics (F,3,4) on a circular orbit around the origin."""
    ang = np.deg2rad(arc_deg) * np.arange(n_frames) / max(n_frames, 1)
    ext = np.empty((n_frames, 3, 4))
    for i, a in enumerate(ang):
        C = (radius * np.cos(a), height, radius * np.sin(a))
        ext[i] = look_at_extrinsic(C)
    return ext


def rodrigues(rvec):
    """Axis-angle -> 3x3 (the mathematical Rodrigues formula; no OpenCV involved)."""
    r = np.asarray(rvec, float).reshape(3)
    th = np.linalg.norm(r)
    if th == 0.0:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def make_ba_problem(n_frames, n_points, track_len=6, seed=1, width=1920, height=1080,
                    obs_sigma=0.5, point_sigma=0.02, pose_sigma=0.002, arc_deg=None, K=None):
    """Bundle-adjustment micro workload.

    P points uniform in a 4^3 cube, each observed in `track_len` consecutive frames of an orbit,
    observation noise sigma px, initial point noise, small pose noise.  Observations are emitted
    point-major with frames ascending inside a point — the order `managePoints` produces
    (/root/reference/processor.py:280-289).

    Returns dict(ext[F,3,4], K, pts0[P,3], obs[O,2], fi[O], pi[O], pts_gt, ext_gt).
    """
    rng = np.random.default_rng(seed)
    F, P, L = n_frames, n_points, min(track_len, n_frames)
    if arc_deg is None:
        arc_deg = min(360.0, 0.72 * F)
    if K is None:
        K = default_K(width, height)
    ext_gt = orbit_cameras(F, arc_deg=arc_deg)
    pts_gt = rng.uniform(-2.0, 2.0, size=(P, 3))
    start = rng.integers(0, F - L + 1, size=P)
    fi = (start[:, None] + np.arange(L)[None, :]).reshape(-1).astype(np.int64)
    pi = np.repeat(np.arange(P, dtype=np.int64), L)
    Xc = np.einsum("oij,oj->oi", ext_gt[fi, :, :3], pts_gt[pi]) + ext_gt[fi, :, 3]
    u = Xc @ K.T
    obs = u[:, :2] / u[:, 2:3] + rng.normal(0.0, obs_sigma, size=(fi.size, 2))
    pts0 = pts_gt + rng.normal(0.0, point_sigma, size=(P, 3))
    ext = ext_gt.copy()
    for f in range(F):
        dR = rodrigues(rng.normal(0.0, pose_sigma, 3))
        ext[f, :, :3] = dR @ ext_gt[f, :, :3]
        ext[f, :, 3] = ext_gt[f, :, 3] + rng.normal(0.0, pose_sigma * 5, 3)
    return dict(ext=ext, K=K, pts0=pts0, obs=obs, fi=fi, pi=pi, pts_gt=pts_gt, ext_gt=ext_gt)


def random_descriptors(n, seed=0, flip_p=0.05, match_frac=0.7):
    """Planted-match descriptor pair (SURVEY.md §8d C2 recipe).

    q: random [n,32] u8.  t: a permutation of q where `match_frac` of the rows get
    Binomial(256, flip_p) bit flips and the rest are replaced by fresh random rows.
    Returns (q, t, perm) with t[i] derived from q[perm[i]].
    """
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    perm = rng.permutation(n)
    t = q[perm].copy()
    fresh = rng.random(n) >= match_frac
    flips = rng.random((n, 256)) < flip_p
    t ^= np.packbits(flips, axis=1, bitorder="little")
    t[fresh] = rng.integers(0, 256, size=(int(fresh.sum()), 32), dtype=np.uint8)
    return q, t, perm


# ----------------------------------------------------------------------------------------------
# Procedural scene: a textured ellipsoid standing on a textured ground plane.
# ----------------------------------------------------------------------------------------------

def make_texture(size=2048, n_shapes=6000, seed=7):
    """u8 texture full of corners: random axis-aligned rectangles and discs of random grey."""
    rng = np.random.default_rng(seed)
    tex = np.full((size, size), 128, np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    for _ in range(n_shapes):
        w, h = rng.integers(6, 64, 2)
        x0, y0 = rng.integers(0, size - 64, 2)
        g = np.uint8(rng.integers(0, 256))
        if rng.random() < 0.6:
            tex[y0:y0 + h, x0:x0 + w] = g
        else:
            r = min(w, h) // 2
            m = (xx - r) ** 2 + (yy - r) ** 2 <= r * r
            sub = tex[y0:y0 + 64, x0:x0 + 64]
            sub[m] = g
    return tex


def render_frame(tex, ext, K, width, height, radii=(2.0, 1.4, 2.0), ground_y=1.4, xp=np):
    """Render one grey frame: ray-cast an ellipsoid (centre origin) and the plane y=ground_y.

    World y points down (camera convention of look_at_extrinsic), so the ground is at +y.
    `xp` is numpy or torch-like (only numpy is used in tests; bench renders with torch on device
    through `render_orbit_frames_torch`).
    """
    R = ext[:, :3]
    t = ext[:, 3]
    C = -R.T @ t
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
    d = d_cam @ R  # world direction (R^T d_cam)
    inv = 1.0 / np.asarray(radii)
    o_s, d_s = C * inv, d * inv
    a = (d_s * d_s).sum(-1)
    b = 2.0 * (d_s * o_s).sum(-1)
    c = (o_s * o_s).sum() - 1.0
    disc = b * b - 4 * a * c
    hit_e = disc > 0
    te = np.where(hit_e, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    te = np.where(te > 0, te, np.inf)
    tg = np.where(d[..., 1] > 1e-9, (ground_y - C[1]) / np.where(d[..., 1] > 1e-9, d[..., 1], 1.0), np.inf)
    tg = np.where(tg > 0, tg, np.inf)
    use_e = te <= tg
    tt = np.minimum(te, tg)
    valid = np.isfinite(tt)
    X = C + d * np.where(valid, tt, 0.0)[..., None]
    S = tex.shape[0]
    # texture coordinates: ellipsoid by (azimuth, polar), ground by (x, z)
    az = np.arctan2(X[..., 2] * inv[2], X[..., 0] * inv[0])
    pol = np.arccos(np.clip(X[..., 1] * inv[1], -1, 1))
    ue = (az / (2 * np.pi) + 0.5) * (S - 1)
    ve = (pol / np.pi) * (S - 1)
    ug = (X[..., 0] / 24.0 + 0.5) * (S - 1)
    vg = (X[..., 2] / 24.0 + 0.5) * (S - 1)
    tu = np.where(use_e, ue, ug)
    tv = np.where(use_e, ve, vg)
    tu = np.clip(tu, 0, S - 1.001)
    tv = np.clip(tv, 0, S - 1.001)
    x0 = np.floor(tu).astype(np.int64)
    y0 = np.floor(tv).astype(np.int64)
    fxw, fyw = tu - x0, tv - y0
    T = tex.astype(np.float64)
    val = (T[y0, x0] * (1 - fxw) * (1 - fyw) + T[y0, x0 + 1] * fxw * (1 - fyw)
           + T[y0 + 1, x0] * (1 - fxw) * fyw + T[y0 + 1, x0 + 1] * fxw * fyw)
    img = np.where(valid, val, 30.0)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def render_orbit_frames(n_frames, width, height, arc_deg=30.0, seed=7, tex_size=1024, K=None):
    """(frames [F,H,W] u8, ext [F,3,4], K) for a seeded orbit clip (numpy, CPU)."""
    if K is None:
        K = default_K(width, height, f=525.0 * width / 640.0)
    tex = make_texture(tex_size, n_shapes=int(6000 * (tex_size / 2048) ** 2) + 500, seed=seed)
    ext = orbit_cameras(n_frames, arc_deg=arc_deg, radius=7.0, height=-2.0)
    frames = np.stack([render_frame(tex, ext[i], K, width, height) for i in range(n_frames)])
    return frames, ext, K


def render_orbit_frames_torch(n_frames, width, height, device, arc_deg=30.0, seed=7, tex_size=2048,
                              K=None, chunk=8):
    """Same scene rendered with torch on `device` (bench input synthesis only; not bit-identical
    to the numpy renderer — the bench hands the *rendered* frames to both the GPU path and the
    CPU baseline, so they still see identical inputs)."""
    import torch
    if K is None:
        K = default_K(width, height, f=525.0 * width / 640.0)
    tex_np = make_texture(tex_size, n_shapes=int(6000 * (tex_size / 2048) ** 2) + 500, seed=seed)
    tex = torch.from_numpy(tex_np).to(device=device, dtype=torch.float32)
    ext = orbit_cameras(n_frames, arc_deg=arc_deg, radius=7.0, height=-2.0)
    S = tex_size
    u, v = torch.meshgrid(torch.arange(width, device=device, dtype=torch.float32),
                          torch.arange(height, device=device, dtype=torch.float32), indexing="xy")
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    d_cam = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], -1)
    radii = torch.tensor([2.0, 1.4, 2.0], device=device)
    inv = 1.0 / radii
    out = torch.empty((n_frames, height, width), dtype=torch.uint8, device=device)
    for i in range(n_frames):
        R = torch.from_numpy(ext[i, :, :3]).to(device=device, dtype=torch.float32)
        t = torch.from_numpy(ext[i, :, 3]).to(device=device, dtype=torch.float32)
        C = -(R.T @ t)
        d = d_cam @ R
        o_s, d_s = C * inv, d * inv
        a = (d_s * d_s).sum(-1)
        b = 2.0 * (d_s * o_s).sum(-1)
        c = (o_s * o_s).sum() - 1.0
        disc = b * b - 4 * a * c
        inf = torch.full_like(a, float("inf"))
        te = torch.where(disc > 0, (-b - torch.sqrt(disc.clamp_min(0))) / (2 * a), inf)
        te = torch.where(te > 0, te, inf)
        dy = d[..., 1]
        tg = torch.where(dy > 1e-9, (1.4 - C[1]) / torch.where(dy > 1e-9, dy, torch.ones_like(dy)), inf)
        tg = torch.where(tg > 0, tg, inf)
        use_e = te <= tg
        tt = torch.minimum(te, tg)
        valid = torch.isfinite(tt)
        X = C + d * torch.where(valid, tt, torch.zeros_like(tt))[..., None]
        az = torch.atan2(X[..., 2] * inv[2], X[..., 0] * inv[0])
        pol = torch.acos((X[..., 1] * inv[1]).clamp(-1, 1))
        tu = torch.where(use_e, (az / (2 * np.pi) + 0.5) * (S - 1), (X[..., 0] / 24.0 + 0.5) * (S - 1))
        tv = torch.where(use_e, (pol / np.pi) * (S - 1), (X[..., 2] / 24.0 + 0.5) * (S - 1))
        tu = tu.clamp(0, S - 1.001)
        tv = tv.clamp(0, S - 1.001)
        x0 = tu.floor().long()
        y0 = tv.floor().long()
        wx, wy = tu - x0, tv - y0
        val = (tex[y0, x0] * (1 - wx) * (1 - wy) + tex[y0, x0 + 1] * wx * (1 - wy)
               + tex[y0 + 1, x0] * (1 - wx) * wy + tex[y0 + 1, x0 + 1] * wx * wy)
        img = torch.where(valid, val, torch.full_like(val, 30.0))
        out[i] = img.round().clamp(0, 255).to(torch.uint8)
    return out, ext, K
