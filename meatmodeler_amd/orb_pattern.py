"""The 256 rBRIEF point pairs used by the ORB descriptor.

OpenCV ships a *learned* table (``bit_pattern_31_`` in features2d/src/orb.cpp, reached by the reference through
``cv2.ORB_create`` at /root/reference/processor.py:308).  That table is data inside a third-party library that is
absent offline and cannot be regenerated, so this build ships its own fixed table drawn once from the BRIEF
"G II" recipe (isotropic Gaussian, sigma = patch/5) with a fixed seed, clipped to the radius-15 disc so that every
rotation of a point stays inside the 33x33 blurred patch.  Descriptors are therefore NOT interchangeable with
OpenCV's; matching only ever compares descriptors produced by this table.
"""
import numpy as np

_PATTERN = None


def brief_pattern(seed=0x0B51EF):
    """[256, 4] int8 rows (x0, y0, x1, y1)."""
    global _PATTERN
    if _PATTERN is None:
        rng = np.random.default_rng(seed)
        pts = []
        while len(pts) < 512:
            p = np.rint(rng.normal(0.0, 31.0 / 5.0, 2)).astype(int)
            if p[0] * p[0] + p[1] * p[1] <= 15 * 15:
                pts.append(p)
        pat = np.array(pts, np.int8).reshape(256, 4)
        # a pair of identical points would give a constant bit: nudge the second point
        same = (pat[:, 0] == pat[:, 2]) & (pat[:, 1] == pat[:, 3])
        pat[same, 2] = np.where(pat[same, 2] < 10, pat[same, 2] + 1, pat[same, 2] - 1)
        _PATTERN = pat
    return _PATTERN
