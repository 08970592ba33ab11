"""ctypes binding of libmeatmodeler_hip.so (C ABI: include/meatmodeler.h).

There is NO CPU fallback: if the shared library is missing the import of any product module fails
loudly, and every compute call needs a HIP device.  Build with ``make -C meatmodeler_amd/csrc`` (or
``python -c 'import __graft_entry__ as g; g.build()'``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmeatmodeler_hip.so")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)
c_i8p = C.POINTER(C.c_int8)
vp = C.c_void_p


class MMError(RuntimeError):
    pass


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
                ("fast_threshold", C.c_int32), ("scale_factor", C.c_float), ("reserved", C.c_int32)]


class BAProblem(C.Structure):
    _fields_ = [("F", C.c_int32), ("P", C.c_int32), ("O", C.c_int64), ("K", vp), ("fi", vp), ("pi", vp),
                ("obs", vp), ("pt_ptr", vp), ("pt_obs", vp), ("cam_ptr", vp), ("cam_obs", vp),
                ("cam_span", C.c_int32), ("reserved", C.c_int32), ("n_seg", C.c_int64), ("seg_ids", vp),
                ("seg_chunk_ptr", vp), ("n_chunks", C.c_int64), ("chunk_seg", vp), ("chunk_begin", vp),
                ("chunk_end", vp), ("pair_o", vp), ("pair_o2", vp), ("pair_p", vp)]


class TrfParams(C.Structure):
    _fields_ = [("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double), ("min_damping", C.c_double),
                ("max_nfev", C.c_int64)]


class TrfRow(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("nfev", C.c_int32), ("cost", C.c_double), ("reduction", C.c_double),
                ("step_norm", C.c_double), ("optimality", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int64)


class Dist(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("half_bandwidth", C.c_int32), ("band_exchange", C.c_int32),
                ("allreduce", ALLREDUCE_FN), ("user", vp)]


class TrfReport(C.Structure):
    _fields_ = [("cost0", C.c_double), ("cost", C.c_double), ("optimality", C.c_double), ("min_damping", C.c_double),
                ("nfev", C.c_int32), ("njev", C.c_int32), ("status", C.c_int32), ("iterations", C.c_int32),
                ("log_rows", C.c_int32), ("chol_fallbacks", C.c_int32), ("collectives", C.c_int32), ("reserved", C.c_int32)]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header.
SIGNATURES = {
    "mm_abi_version": (C.c_int, []),
    "mm_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "mm_ctx_destroy": (None, [vp]),
    "mm_ctx_control": (C.c_longlong, [vp, C.c_int, C.c_longlong]),
    "mm_last_error": (C.c_char_p, [vp]),
    "mm_ctx_sync": (C.c_int, [vp]),
    "mm_timer_create": (C.c_int, [vp, C.POINTER(vp)]),
    "mm_timer_start": (C.c_int, [vp, vp]),
    "mm_timer_stop": (C.c_int, [vp, vp]),
    "mm_timer_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "mm_timer_destroy": (None, [vp, vp]),
    "mm_profile_enable": (C.c_int, [vp, C.c_int]),
    "mm_profile_report": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "mm_profile_select": (C.c_int, [vp, C.c_char_p]),
    "mm_bf_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mm_bf_knn2_batched": (C.c_int, [vp, vp, vp, C.c_int, C.c_size_t, vp, vp, C.c_int, C.c_size_t, C.c_int, vp, vp,
                                     vp, C.c_size_t]),
    "mm_bf_knn2_hamming": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_size_t]),
    "mm_ratio_filter_batched": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp, vp]),
    "mm_orb_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.POINTER(OrbParams)]),
    "mm_orb_detect_compute": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrbParams), vp, vp,
                                        C.c_size_t, vp, vp, vp, vp, vp, vp]),
    "mm_orb_level_sizes": (C.c_int, [C.c_int, C.c_int, C.POINTER(OrbParams), c_i32p, c_i32p, c_i32p, c_f32p]),
    "mm_link_tracks_clip": (C.c_int64, [C.c_int, C.c_int, c_i32p, c_f32p, C.c_int, c_i32p, c_i32p, C.c_int64,
                                        C.c_int64, c_i64p, c_i32p, c_i32p, c_i64p]),
    "mm_link_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mm_link_tracks_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp]),
    "mm_flatten_offsets": (C.c_int, [vp, vp, vp, C.c_int64, vp]),
    "mm_flatten_tracks": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int64, vp, C.c_int64, C.c_int, vp, vp, vp]),
    "mm_ba_build_pairs": (C.c_int64, [C.c_int, C.c_int, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, C.c_int,
                                      c_i64p, c_i32p, c_i32p, C.c_int64]),
    "mm_ba_build_index": (C.c_int, [C.c_int, C.c_int, C.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p, c_i32p]),
    "mm_triangulate_dlt": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int64, vp]),
    "mm_ba_residual": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, C.c_size_t]),
    "mm_ba_jacobian": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp]),
    "mm_ba_normal_eq": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, vp]),
    "mm_ba_jvp": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp]),
    "mm_ba_schur": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t]),
    "mm_ba_schur_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem)]),
    "mm_ba_pairs_count": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp]),
    "mm_ba_pairs_emit": (C.c_int, [vp, C.POINTER(BAProblem), vp, C.c_int, vp, vp, vp]),
    "mm_ba_schur_solve": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp,
                                    C.c_size_t, vp, C.c_size_t, C.c_int, C.c_int, c_i64p, c_i64p]),
    "mm_multi_dot_workspace_bytes": (C.c_size_t, []),
    "mm_multi_dot": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int64, C.c_int64, vp, vp, C.c_size_t]),
    "mm_trf_fused": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_double, C.c_double, C.c_int64,
                               C.c_int64, vp, vp, C.c_size_t]),
    "mm_trf_damping": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, vp]),
    "mm_ba_backsub": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, vp, vp, C.c_size_t]),
    "mm_ba_backsub_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem)]),
    "mm_chol_workspace_bytes": (C.c_size_t, [C.c_int]),
    "mm_chol_solve": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_size_t]),
    "mm_ba_jvp_dots_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem)]),
    "mm_ba_jvp_dots": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t]),
    "mm_ba_scale_update": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]),
    "mm_ba_damp": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "mm_trf_step2d": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, vp]),
    "mm_pyr_down": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]),
    "mm_lk_track": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, vp, vp, vp]),
    "mm_min_eig": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mm_corner_candidates": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, vp, vp, vp, C.c_int, vp]),
    "mm_gftt_select": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, vp, C.c_int]),
    "mm_contrast_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mm_increase_contrast": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_double, C.c_int, C.c_int, vp, vp, vp,
                                       C.c_size_t]),
    "mm_bgr_to_grey": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "mm_write_ply": (C.c_int, [C.c_char_p, vp, C.c_int64]),
    "mm_chol_solve_sym": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_size_t]),
    "mm_ba_trf_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem)]),
    "mm_ba_trf": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, C.POINTER(TrfParams), C.POINTER(TrfReport),
                            C.POINTER(TrfRow), C.c_int, vp, C.c_size_t]),
    "mm_ba_trf_batched_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem)]),
    "mm_ba_trf_batched": (C.c_int, [vp, C.c_int, C.POINTER(C.POINTER(BAProblem)), C.POINTER(vp), C.POINTER(vp),
                                    C.POINTER(TrfParams), C.POINTER(TrfReport), C.POINTER(vp), C.POINTER(C.c_size_t), c_i32p]),
    "mm_ba_trf_dist_workspace_bytes": (C.c_size_t, [C.POINTER(BAProblem), C.c_int]),
    "mm_ba_trf_dist": (C.c_int, [vp, C.POINTER(BAProblem), vp, vp, C.POINTER(TrfParams), C.POINTER(TrfReport),
                                 C.POINTER(TrfRow), C.c_int, vp, C.c_size_t, C.POINTER(Dist)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(make -C meatmodeler_amd/csrc).  meatmodeler_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


class Context:
    """One mm_ctx per (device, stream).  Requires a HIP device — raises MMError otherwise."""

    def __init__(self, device=None, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise MMError("meatmodeler_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                          "there is no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = vp()
        rc = lib.mm_ctx_create(self.device.index, vp(self.stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise MMError(f"mm_ctx_create failed ({rc})")
        self.h = h

    def check(self, rc, what=""):
        if rc != 0:
            msg = lib.mm_last_error(self.h)
            raise MMError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def sync(self):
        self.check(lib.mm_ctx_sync(self.h), "mm_ctx_sync")

    CTL_CHOL_FORCE_ABANDON, CTL_CHOL_LAST_PATH, CTL_CHOL_RESERVED, CTL_CU_COUNT, CTL_CHOL_AVOID_FUSED = 1, 2, 3, 4, 5
    CTL_LINK_LAST_VARIANT = 6
    CTL_BATCH_LAST = 7

    def control(self, what, value=0):
        """mm_ctx_control: knobs / queries of the context (see include/meatmodeler.h)."""
        r = int(lib.mm_ctx_control(self.h, int(what), int(value)))
        if r < -1 or (r == -1 and what not in (self.CTL_CHOL_LAST_PATH, self.CTL_LINK_LAST_VARIANT, self.CTL_BATCH_LAST)):
            self.check(r, "mm_ctx_control")
        return r

    def profile(self, level, only=None):
        """0 off, 1 every launch, 2 launches of >= 64 workgroups only, 3 launches of the kernel named `only`."""
        if only is not None:
            self.check(lib.mm_profile_select(self.h, only.encode()), "mm_profile_select")
        self.check(lib.mm_profile_enable(self.h, int(level)), "mm_profile_enable")

    def profile_report(self):
        """{kernel name: (launches, total_ms)} of the launches recorded since profile(True)."""
        buf = C.create_string_buffer(1 << 16)
        n = lib.mm_profile_report(self.h, buf, len(buf))
        if n < 0:
            self.check(n, "mm_profile_report")
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.mm_ctx_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Timer:
    """HIP events on the context's stream (bench.py roofline leg)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.t = vp()
        ctx.check(lib.mm_timer_create(ctx.h, C.byref(self.t)), "mm_timer_create")

    def start(self):
        self.ctx.check(lib.mm_timer_start(self.ctx.h, self.t), "mm_timer_start")

    def stop(self):
        self.ctx.check(lib.mm_timer_stop(self.ctx.h, self.t), "mm_timer_stop")

    def elapsed_ms(self):
        ms = C.c_float()
        self.ctx.check(lib.mm_timer_elapsed_ms(self.ctx.h, self.t, C.byref(ms)), "mm_timer_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib.mm_timer_destroy(self.ctx.h, self.t)
        except Exception:
            pass


_default_ctx = {}


def default_context():
    import torch
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    c = _default_ctx.get(key)
    if c is None:
        c = _default_ctx[key] = Context()
    return c
