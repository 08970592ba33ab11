"""CPU: a SECOND SOURCE for the parity-unpinned C oracles (oracle/orb_oracle.c, oracle/frame_oracle.c,
ba_oracle.triangulate_dlt).

OpenCV is absent (the reference reaches these stages through cv2: /root/reference/processor.py:79,104,129,132-137,259)
and the reference holds no fixture for them, so the C oracles cannot be pinned.  What CAN be done without OpenCV is to
write every stage a second time, straight from its published definition, in a different formulation (whole-image NumPy,
brute force, exact rationals) and require equality with the C oracle: a misreading shared by the oracle and the HIP
kernels (circle order, arc length, Harris window, disc table, BRIEF bit order, blur weights, tie rules) would have to be
made a third time, independently, to pass.  The definitions are the ones DESIGN.md §3 / §3b state.
"""
from fractions import Fraction

import numpy as np
import pytest

from meatmodeler_amd.orb_pattern import brief_pattern
from oracle import ba_oracle as bo
from oracle import frame_oracle as fo
from oracle import orb_oracle as oo


# ------------------------------------------------------------------------------------------------ test images
def _texture(h, w, seed):
    """Blobs, bars and noise: plenty of FAST corners at several contrasts, flat areas, saturated pixels."""
    rng = np.random.default_rng(seed)
    img = rng.integers(90, 110, size=(h, w)).astype(np.int32)
    for _ in range(60):
        y, x = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
        hh, ww = int(rng.integers(3, 14)), int(rng.integers(3, 14))
        img[y:y + hh, x:x + ww] = int(rng.integers(0, 256))
    yy, xx = np.mgrid[0:h, 0:w]
    img += (20 * np.sin(xx / 5.0) * np.cos(yy / 7.0)).astype(np.int32)
    return np.clip(img, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ FAST-9/16
# Bresenham circle of radius 3, clockwise from 12 o'clock (Rosten & Drummond, fig. 1); (dx, dy)
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
          (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]


def _segment_test(img, t):
    """Boolean map: nine contiguous circle pixels all brighter than p + t or all darker than p - t (the definition)."""
    h, w = img.shape
    p = img[3:h - 3, 3:w - 3].astype(np.int32)
    ring = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in CIRCLE])
    out = np.zeros_like(p, bool)
    for flags in (ring > p + t, ring < p - t):
        for start in range(16):
            run = np.ones_like(out)
            for j in range(9):
                run &= flags[(start + j) % 16]
            out |= run
    full = np.zeros((h, w), bool)
    full[3:h - 3, 3:w - 3] = out
    return full


def _fast_score_bruteforce(img, t0=20):
    """score = the largest threshold at which the pixel still passes the segment test (OpenCV's cornerScore), for
    pixels that pass at t0; found by trying every threshold, no monotonicity assumed."""
    score = np.zeros(img.shape, np.int32)
    passes0 = _segment_test(img, t0)
    for t in range(t0, 256):
        m = _segment_test(img, t) & passes0
        if not m.any():
            continue
        score[m] = np.maximum(score[m], t)
    return score


def test_fast_score_is_largest_passing_threshold():
    for seed, (h, w) in enumerate([(48, 64), (40, 40)]):
        img = _texture(h, w, seed)
        want = _fast_score_bruteforce(img)
        got = oo.fast_score_map(img, 20).astype(np.int32)
        assert (want > 0).sum() > 20
        assert np.array_equal(got, want)


# ------------------------------------------------------------------------------------------------ Harris
def _harris_map(img):
    """(a b - c^2) - 0.04 (a + b)^2 over the 7x7 block of 3x3 Sobel derivatives, times 25 to stay in integers."""
    I = img.astype(np.int64)
    h, w = I.shape
    gx = np.zeros_like(I)
    gy = np.zeros_like(I)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    for dy in range(3):
        for dx in range(3):
            gx[1:h - 1, 1:w - 1] += kx[dy, dx] * I[dy:h - 2 + dy, dx:w - 2 + dx]
            gy[1:h - 1, 1:w - 1] += kx.T[dy, dx] * I[dy:h - 2 + dy, dx:w - 2 + dx]

    def box7(a):
        s = np.zeros_like(a)
        for dy in range(-3, 4):
            for dx in range(-3, 4):
                s[4:h - 4, 4:w - 4] += a[4 + dy:h - 4 + dy, 4 + dx:w - 4 + dx]
        return s
    a, b, c = box7(gx * gx), box7(gy * gy), box7(gx * gy)
    return 25 * (a * b - c * c) - (a + b) ** 2


def test_harris_response_definition():
    img = _texture(40, 56, 3)
    H = _harris_map(img)
    for y in range(4, 36, 3):
        for x in range(4, 52, 5):
            assert oo.harris25_at(img, x, y) == int(H[y, x])


# ------------------------------------------------------------------------------------------------ orientation + BRIEF
# OpenCV's umax for a patch of 31 (orb.cpp; the table every ORB implementation prints): half-widths of the disc rows
UMAX_31 = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
BLUR_TAPS = np.array([18, 33, 49, 56, 49, 33, 18], np.int64)   # 7 taps, sigma 2, sum 256


def test_umax_table():
    assert oo.umax().tolist() == UMAX_31


def _blur_image(img):
    """7x7 Gaussian as an outer product of the integer taps, rounded once: (sum + 2^15) >> 16; valid interior only."""
    I = img.astype(np.int64)
    h, w = I.shape
    acc = np.zeros((h - 6, w - 6), np.int64)
    for i in range(7):
        for j in range(7):
            acc += BLUR_TAPS[i] * BLUR_TAPS[j] * I[i:h - 6 + i, j:w - 6 + j]
    out = np.zeros((h, w), np.int64)
    out[3:h - 3, 3:w - 3] = (acc + 32768) >> 16
    return out


def _describe_np(img, blur, x, y, pattern):
    vv, uu = np.mgrid[-15:16, -15:16]
    disc = np.abs(uu) <= np.array(UMAX_31)[np.abs(vv)]
    patch = img[y - 15:y + 16, x - 15:x + 16].astype(np.int64)
    m10, m01 = int((uu * patch)[disc].sum()), int((vv * patch)[disc].sum())
    norm = np.sqrt(np.float64(m10 * m10 + m01 * m01))
    cs, sn = (np.float64(m10) / norm, np.float64(m01) / norm) if norm > 0 else (1.0, 0.0)
    bits = np.zeros(256, np.uint8)
    pat = pattern.astype(np.float64)
    for k in range(256):
        vals = []
        for e in range(2):
            px, py = pat[k, 2 * e], pat[k, 2 * e + 1]
            ix, iy = int(np.rint(px * cs - py * sn)), int(np.rint(px * sn + py * cs))
            vals.append(blur[y + iy, x + ix])
        bits[k] = vals[0] < vals[1]
    # bit j of byte k = test 8k + j  -> little-endian bit order inside a byte
    return np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").ravel(), m10, m01


def test_blur_moments_and_brief_bits():
    img = _texture(96, 96, 5)
    blur = _blur_image(img)
    pat = brief_pattern()
    for (x, y) in [(40, 40), (31, 31), (64, 33), (50, 62), (33, 64)]:
        assert oo.blurred_at(img, x, y) == int(blur[y, x])
        d, m10, m01 = oo.describe(img, x, y, pat)
        dn, m10n, m01n = _describe_np(img, blur, x, y, pat)
        assert (m10, m01) == (m10n, m01n)
        assert np.array_equal(d, dn)


# ------------------------------------------------------------------------------------------------ pyramid
def _resize_exact(src, wd, hd):
    """Bilinear at the sample position (d + 1/2) ns/nd - 1/2 (clamped to the image), weights rounded to 11 bits,
    one final rounding: exact rationals, pixel by pixel."""
    hs, ws = src.shape

    def axis(d, nd, ns):
        s = (Fraction(2 * d + 1, 2) * ns) / nd - Fraction(1, 2)
        if s < 0:
            s = Fraction(0)
        i0 = int(s)   # floor for s >= 0
        wq = int((s - i0) * 2048 + Fraction(1, 2))   # round half up
        if i0 >= ns - 1:
            i0, wq = ns - 1, 0
        return i0, min(i0 + 1, ns - 1), wq
    out = np.zeros((hd, wd), np.uint8)
    xs = [axis(x, wd, ws) for x in range(wd)]
    for y in range(hd):
        y0, y1, wy = axis(y, hd, hs)
        for x in range(wd):
            x0, x1, wx = xs[x]
            top = int(src[y0, x0]) * (2048 - wx) + int(src[y0, x1]) * wx
            bot = int(src[y1, x0]) * (2048 - wx) + int(src[y1, x1]) * wx
            out[y, x] = (top * (2048 - wy) + bot * wy + (1 << 21)) >> 22
    return out


def test_bilinear_resize_definition():
    src = _texture(60, 83, 7)
    for (wd, hd) in [(69, 50), (58, 42), (83, 60), (41, 30)]:
        assert np.array_equal(oo.resize(src, wd, hd), _resize_exact(src, wd, hd))


def test_level_geometry_definition():
    """scale_l = float32(1.2^l); size = round(dim / scale_l); n_l geometric with ratio 1/1.2, remainder on the last."""
    H, W, nf = 1080, 1920, 4000
    w, h, n, s = oo.level_sizes(H, W, nf)
    f = np.float32(1.0 / np.float32(1.2))
    nd = np.float32(nf) * (np.float32(1) - f) / (np.float32(1) - np.float32(float(f) ** 8))
    want_n = []
    for l in range(7):
        want_n.append(int(np.rint(nd)))
        nd = np.float32(nd * f)
    want_n.append(nf - sum(want_n))
    assert n.tolist() == want_n and sum(want_n) == nf
    for l in range(8):
        sc = np.float32(float(np.float32(1.2)) ** l)
        assert s[l] == sc
        assert (w[l], h[l]) == ((W, H) if l == 0 else (int(np.rint(np.float32(W) / sc)), int(np.rint(np.float32(H) / sc))))


# ------------------------------------------------------------------------------------------------ the whole ORB chain
def _detect_compute_np(img, nfeatures, pattern):
    w, h, n, scale = oo.level_sizes(img.shape[0], img.shape[1], nfeatures)   # (checked on its own above)
    cur, out = img, []
    for l in range(8):
        if l > 0:
            cur = _resize_exact(cur, int(w[l]), int(h[l]))
        hl, wl = cur.shape
        if wl <= 62 or hl <= 62 or n[l] == 0:
            continue
        score = _fast_score_bruteforce(cur)
        # strict 3x3 non-maximum suppression on the score map, key points at least 31 px from the border
        keep = score > 0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    sh = np.zeros_like(score)
                    sh[max(0, -dy):hl - max(0, dy), max(0, -dx):wl - max(0, dx)] = \
                        score[max(0, dy):hl - max(0, -dy), max(0, dx):wl - max(0, -dx)]
                    keep &= score > sh
        inner = np.zeros_like(keep)
        inner[31:hl - 31, 31:wl - 31] = True
        ys, xs = np.nonzero(keep & inner)
        sc = score[ys, xs]
        order = np.lexsort((xs, ys, -sc))[:2 * n[l]]
        ys, xs = ys[order], xs[order]
        Hm = _harris_map(cur)
        hv = Hm[ys, xs]
        order = np.lexsort((xs, ys, -hv))[:n[l]]
        blur = _blur_image(cur)
        for y, x, hh in zip(ys[order], xs[order], hv[order]):
            d, m10, m01 = _describe_np(cur, blur, int(x), int(y), pattern)
            out.append((l, int(x), int(y), int(hh), m10, m01, d,
                        np.float32(x) * scale[l], np.float32(y) * scale[l]))
    return out


def test_orb_chain_numpy_equals_c_oracle():
    """The full detect + describe chain, re-implemented on whole images in NumPy, equals orb_oracle.c key point by key
    point: level, position, Harris value, moments, 32 descriptor bytes, output order."""
    img = _texture(150, 200, 11)
    pat = brief_pattern()
    for nf in (60, 400):
        got = oo.detect_compute(img, nf, pat)
        want = _detect_compute_np(img, nf, pat)
        assert got["n"] == len(want) and got["n"] > 20
        for i, (l, x, y, hh, m10, m01, d, fx, fy) in enumerate(want):
            assert got["meta"][i, :3].tolist() == [l, x, y]
            assert got["meta"][i, 3] == np.int64(hh).astype(np.int32)     # low 32 bits, as the record stores them
            assert got["mom"][i].tolist() == [m10, m01]
            assert np.array_equal(got["desc"][i], d)
            assert got["xy"][i, 0] == fx and got["xy"][i, 1] == fy
            assert got["resp"][i] == np.float32(hh) * np.float32(1.0 / (25.0 * 7140.0 ** 4))


# ------------------------------------------------------------------------------------------------ matching
def test_hamming_knn2_and_ratio_definition():
    rng = np.random.default_rng(2)
    q = rng.integers(0, 256, size=(70, 32), dtype=np.uint8)
    t = rng.integers(0, 256, size=(90, 32), dtype=np.uint8)
    t[5] = t[40] = q[3]          # exact duplicates: ties go to the lowest train index
    t[17] = q[9]
    t[60, :31] = q[9, :31]
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.int32)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    idx, dist = oo.bf_knn2(q, t)
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.take_along_axis(D, order, 1))
    keep = dist[:, 0] < 0.75 * dist[:, 1]
    pairs = oo.ratio_filter(idx, dist, 0.75)
    assert np.array_equal(pairs, np.stack([np.nonzero(keep)[0], idx[keep, 0]], 1))
    # fewer than two train descriptors: no second neighbour, nothing passes the ratio test
    idx1, dist1 = oo.bf_knn2(q[:4], t[:1])
    assert (idx1[:, 1] == -1).all() and len(oo.ratio_filter(idx1, dist1)) == 0


# ------------------------------------------------------------------------------------------------ triangulation
def test_dlt_by_normal_equations_eigenvector():
    """X = the eigenvector of A^T A with the smallest eigenvalue (same minimiser as the SVD of A), A from the two
    projections; and exact recovery of noise-free points."""
    rng = np.random.default_rng(4)
    n = 40
    K = np.array([[900.0, 0, 320], [0, 900.0, 240], [0, 0, 1]])
    X = rng.uniform(-1, 1, size=(n, 3)) + [0, 0, 6]

    def cam(rx, tx):
        R = bo.rodrigues_matrix(np.array([0.0, rx, 0.0]))
        return K @ np.hstack([R, np.array([[tx], [0.0], [0.0]])])
    P1, P2 = cam(0.0, 0.0), cam(0.05, -0.6)
    Xh = np.hstack([X, np.ones((n, 1))])
    x1 = (P1 @ Xh.T).T
    x2 = (P2 @ Xh.T).T
    x1, x2 = x1[:, :2] / x1[:, 2:], x2[:, :2] / x2[:, 2:]
    x1n, x2n = x1 + rng.normal(0, 0.3, x1.shape), x2 + rng.normal(0, 0.3, x2.shape)
    P1s, P2s = np.broadcast_to(P1, (n, 3, 4)), np.broadcast_to(P2, (n, 3, 4))
    assert np.allclose(bo.triangulate_dlt(P1s, P2s, x1, x2), X, rtol=0, atol=1e-8)
    got = bo.triangulate_dlt(P1s, P2s, x1n, x2n)
    for i in range(n):
        A = np.array([x1n[i, 0] * P1[2] - P1[0], x1n[i, 1] * P1[2] - P1[1],
                      x2n[i, 0] * P2[2] - P2[0], x2n[i, 1] * P2[2] - P2[1]])
        v = np.linalg.eigh(A.T @ A)[1][:, 0]
        assert np.allclose(got[i], v[:3] / v[3], rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------------------------------------ frame front end
def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def test_pyr_down_definition():
    """[1 4 6 4 1]/16 in both directions on the reflect-101 extension, every second pixel, (sum + 128) >> 8."""
    img = _texture(37, 50, 9)
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)
    ho, wo = (h + 1) // 2, (w + 1) // 2
    want = np.zeros((ho, wo), np.int64)
    ys, xs = np.arange(ho) * 2, np.arange(wo) * 2
    for i in range(5):
        for j in range(5):
            yy = _reflect101(ys + i - 2, h)
            xx = _reflect101(xs + j - 2, w)
            want += k[i] * k[j] * img[np.ix_(yy, xx)].astype(np.int64)
    assert np.array_equal(fo.pyr_down(img), ((want + 128) >> 8).astype(np.uint8))


def test_grey_definition():
    """Y = 0.299 R + 0.587 G + 0.114 B in 14-bit fixed point (4899, 9617, 1868), rounded."""
    rng = np.random.default_rng(6)
    bgr = rng.integers(0, 256, size=(20, 30, 3), dtype=np.uint8)
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    assert (round(0.299 * 16384), round(0.587 * 16384), round(0.114 * 16384)) == (4899, 9617, 1868)
    assert np.array_equal(fo.bgr_to_grey(bgr), ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8))


def test_min_eigenvalue_definition():
    """Shi-Tomasi: smaller eigenvalue of the block-summed structure tensor of Sobel derivatives scaled by
    1 / (4 block 255), by np.linalg.eigvalsh pixel by pixel."""
    img = _texture(24, 30, 8)
    h, w = img.shape
    bs = 3
    I = img.astype(np.float64)

    def at(y, x):
        return I[_reflect101(np.asarray(y), h), _reflect101(np.asarray(x), w)]
    e = fo.min_eig(img, bs)
    sc = 1.0 / (4 * bs * 255.0)
    for y in range(0, h, 5):
        for x in range(0, w, 7):
            a = b = c = 0.0
            for dy in range(-(bs // 2), bs // 2 + 1):
                for dx in range(-(bs // 2), bs // 2 + 1):
                    yy, xx = int(_reflect101(np.asarray(y + dy), h)), int(_reflect101(np.asarray(x + dx), w))
                    gx = (at(yy - 1, xx + 1) + 2 * at(yy, xx + 1) + at(yy + 1, xx + 1)) - \
                         (at(yy - 1, xx - 1) + 2 * at(yy, xx - 1) + at(yy + 1, xx - 1))
                    gy = (at(yy + 1, xx - 1) + 2 * at(yy + 1, xx) + at(yy + 1, xx + 1)) - \
                         (at(yy - 1, xx - 1) + 2 * at(yy - 1, xx) + at(yy - 1, xx + 1))
                    a += gx * gx
                    b += gx * gy
                    c += gy * gy
            lam = np.linalg.eigvalsh(np.array([[a, b], [b, c]]) * sc * sc)[0]
            assert e[y, x] == pytest.approx(lam, rel=1e-9, abs=1e-12)


def test_lk_recovers_a_known_translation():
    """A smooth image shifted by whole pixels: pyramidal LK must return the shift (status 1, sub-0.05 px), the one
    property of cv2.calcOpticalFlowPyrLK that needs no OpenCV to state."""
    yy, xx = np.mgrid[0:160, 0:200].astype(np.float64)
    base = 128 + 50 * np.sin(xx / 9.0) * np.cos(yy / 11.0) + 40 * np.sin((xx + 2 * yy) / 17.0)
    prev = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    dx, dy = 3, -2
    nxt = np.roll(np.roll(prev, dy, 0), dx, 1)
    pts = np.array([[60, 50], [100, 80], [140, 100], [80, 110]], np.float32)
    out, st, err = fo.lk_track(prev, nxt, pts)
    assert st.tolist() == [1, 1, 1, 1]
    assert np.abs(out - (pts + [dx, dy])).max() < 0.05
