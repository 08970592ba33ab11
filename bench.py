#!/usr/bin/env python3
"""bench.py — the hot path (detect -> match -> link -> triangulate -> bundle adjust) on a synthetic clip.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2] (the configuration the metric is quoted on): 500 frames 1080p, 4000 ORB key points
per frame, consecutive-pair BF Hamming matching, track linking, 2-view triangulation, full BA.  A "step" is one pass of
that path over the whole clip with the frames already resident in HBM.  With N > 1 the SAME clip is sharded over the
ranks (strong scaling): frame pairs for detect/match, points for BA, one RCCL all-reduce of the camera-side blocks
per trust-region iteration.

One JSON line on rank 0.  `value` = descriptor-match pairs/s over the match stage of the timed steps (first quantity of
the metric); `ba_residuals_per_s`, `frames_per_s` and `stage_ms` are reported beside it; `roofline` describes the kernel
with the largest share of device time inside the timed steps (per-launch HIP events on the launch stream, see
mm_profile_* in include/meatmodeler.h); `cpu_baseline` times the CPU oracle on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TLOPS = 78.6         # 256 CU x 4 SIMD-32 x 2.4 GHz (157.3 TF FP32 vector / 2)
MFMA_F64_PEAK_TF = 78.6        # v_mfma_f64_16x16x4: 2048 flop / 64 clk / SIMD
MFMA_FP4_PEAK_PF = 10.0        # MI355X_MICROARCH.md: FP4 / FP6 ~10 PF dense (block-scaled f8f6f4 MFMA)
MFMA_FP4_MEASURED_PF = 6.43   # the same instruction on random +1 / -1 operands, every SIMD busy, >= 10 ms (profiles/r03_mfma_rate.txt)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--nfeatures", type=int, default=4000)
    ap.add_argument("--batch", type=int, default=512,
                    help="frames per ORB launch set (the pyramid workspace is ~4.5 MB per frame; one set per clip avoids "
                         "16 x 14 launch tails: detect 34.6 -> 28.7 ms)")
    ap.add_argument("--arc", type=float, default=None, help="orbit arc in degrees (default 0.72 deg/frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--ba-window", type=int, default=0,
                    help="sliding-window BA instead of the global one (SURVEY 8(f)-2): keyframes per window (1 GPU only)")
    ap.add_argument("--ba-stride", type=int, default=0, help="keyframes between windows (default: window / 2)")
    ap.add_argument("--ba-order", default=None, choices=("sequential", "wavefront"),
                    help="window schedule (default: wavefront -- the independent windows of a pass spread over the ranks and "
                         "over --ba-streams streams per GPU; sequential = every window starts from its predecessor's result)")
    ap.add_argument("--ba-window-max-nfev", type=int, default=0,
                    help="evaluation budget per window of the sliding-window adjustment (0: SciPy's default, 100 n).  The window "
                         "trajectories on this outlier-laden synthetic clip are chaotic: the occasional window crawls for thousands of "
                         "evaluations and then IS the stage (it runs alone at the end of its pass)")
    ap.add_argument("--ba-streams", type=int, default=8,
                    help="wavefront schedule: windows in flight per GPU (one HIP stream + host thread each)")
    ap.add_argument("--ba-batched", type=int, default=0,
                    help="wavefront schedule: 1 = all windows of a pass in ONE lock-step solve (mm_ba_trf_batched), 0 = one solve "
                         "per window on --ba-streams streams (default: at the C5 window shape -- 75 k points per window -- both are "
                         "bound by the same total kernel work, 912 vs 927 ms; the lock-step solve wins on smaller problems)")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--no-profile", action="store_true", help="no per-launch HIP events at all (no kernel table, no roofline)")
    ap.add_argument("--profile-timed", choices=("dominant", "big"), default="dominant",
                    help="HIP events inside the TIMED steps: around the launches of the dominant kernel only (default; every "
                         "bracketed launch costs the stream a bubble) or around every launch of >= 64 workgroups (rounds 1-3: "
                         "inflates a BA iteration by ~10 %%)")
    return ap.parse_args()


def algorithmic_work(name, cfg):
    """(bytes, lane_ops, mfma_flops) per STEP for kernel `name` — the per-unit figures of DESIGN.md x the units."""
    F, N = cfg["frames_local"], cfg["nfeatures"]
    lv = cfg["levels"]           # list of (w, h)
    px = [w * h for w, h in lv]
    O, P, Fc = cfg.get("n_obs_local", 0), cfg.get("n_points_local", 0), cfg["frames"]
    it = cfg.get("ba_iters", 0)   # Jacobian evaluations
    nfev = cfg.get("ba_nfev", 0)
    kp = cfg.get("kp_total_local", F * N)
    if name.startswith("bf_knn2_"):
        pairs = cfg["pair_evals_local"]
        # 8 v_xor_b32 + 8 v_bcnt_u32_b32 = 16 lane-ops per descriptor pair is the unit of the roof
        return cfg["pairs_local"] * (32 * 2 * N + 16 * N), pairs * 16.0, 0
    if name == "orb_fast_kernel":
        return F * (sum(px) + 4 * cfg.get("cand_per_frame", 0)), F * sum(px) * 80, 0
    if name == "orb_resize_kernel":
        return F * sum(px[i - 1] + px[i] for i in range(1, len(px))), 0, 0
    if name == "orb_describe_kernel":
        return kp * (39 * 39 + 32 + 8 + 16), 0, 0
    if name == "orb_harris_kernel":
        return F * (4 * cfg.get("cand_per_frame", 0)) + 2 * kp * (81 + 12), 0, 0
    if name == "orb_rank_kernel":
        return 2 * kp * 12 + kp * 32, 0, 0
    if name == "orb_select_kernel":
        return F * 4 * 4 * cfg.get("cand_per_frame", 0), 0, 0
    if name == "ba_residual_kernel":
        return nfev * O * 48, 0, 0                      # idx 8 + obs 16 + point 24 (cameras stay in L2)
    if name in ("ba_point_blocks_kernel", "ba_backsub_kernel"):
        return it * (O * 28 + P * (24 + 72)), 0, 0
    if name == "ba_normal_eq_kernel":      # point blocks + camera blocks in one launch
        return it * (O * 28 + P * (24 + 72) + O * (4 + 4 + 16 + 24) + Fc * 42 * 8), 0, 0
    if name == "ba_camera_blocks_kernel":
        return it * (O * (4 + 4 + 16 + 24) + Fc * 42 * 8), 0, 0
    if name == "ba_jvp_kernel":
        return 2 * it * O * (48 + 16 + 24), 0, 0
    if name == "schur_pairs_kernel":
        # per co-observation pair: point index 4 B + point 24 B + C^-1 48 B gathered; ~254 f64 instructions (two lean
        # evaluations of 64, Z 18, the 2 x 2 middle factor 12, T 24, the block's 72 multiply-adds)
        return it * cfg.get("n_pairs", 0) * (4 + 24 + 48), it * cfg.get("n_pairs", 0) * 254.0, 0
    if name == "schur_init_kernel":
        return it * (6 * Fc) ** 2 * 8, 0, 0
    if name == "chol_update_kernel":     # band only: per block column ~ bwb (bwb + 1) / 2 tile products of 2 * 64^3 flop
        n = 6 * Fc
        bwb = min((6 * cfg.get("cam_span", Fc) + 5 + 63) // 64, (n + 63) // 64)
        return 0, 0, it * ((n + 63) // 64) * bwb * (bwb + 1) / 2 * 2 * 64 ** 3
    if name == "chol_panel_kernel":
        n = 6 * Fc
        bwb = min((6 * cfg.get("cam_span", Fc) + 5 + 63) // 64, (n + 63) // 64)
        return 0, 0, it * ((n + 63) // 64) * bwb * 2 * 64 ** 3
    if name in ("chol_band_fused_kernel", "chol_band_bwd_kernel"):   # per LAUNCH (table() multiplies by the launch count)
        n = 6 * Fc
        nblk = (n + 63) // 64
        bwb = min((6 * cfg.get("cam_span", Fc) + 5 + 63) // 64, nblk)
        if name == "chol_band_bwd_kernel":   # the band of L and the diagonal-block inverses, read once
            return nblk * (bwb + 1) * 64 * 64 * 8, 0, 0
        # per block column: 64^3/3 (factor) + 64^3/3 (inverse) + bwb solves of 64^3 + bwb (bwb + 1) / 2 products of 2 64^3
        return 0, 0, nblk * 64 ** 3 * (2.0 / 3.0 + bwb + bwb * (bwb + 1))
    if name == "chol_diag_kernel":       # 64x64 Cholesky + triangular inverse per block column
        return 0, 0, it * ((6 * Fc + 63) // 64) * 2 * 64 ** 3 / 3.0
    return None, 0, 0


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: the parent starts N fresh ranks itself (one per GPU) through
    torch.distributed.run and relays their output; rank 0 prints the JSON line.  The parent never touches HIP
    (torch.cuda.device_count() does not initialise the device) and never re-execs: children are ordinary subprocesses."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    backend = os.environ.get("MM_DIST_BACKEND", "nccl")
    if ndev < a.gpus and backend != "gloo":
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible; one rank per GPU is required "
                         f"(MM_DIST_BACKEND=gloo allows a functional rehearsal with ranks sharing a GPU)\n")
        return 2
    if ndev < 1:
        sys.stderr.write("bench.py: no GPU visible\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)     # (rehearsal: several ranks may share one GPU with MM_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1
    if use_dist:
        backend = os.environ.get("MM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from meatmodeler_amd import synth, ops
    from meatmodeler_amd._lib import default_context
    from meatmodeler_amd.pipeline import ClipPipeline

    ctx = default_context()
    F, H, W, N = a.frames, a.height, a.width, a.nfeatures
    arc = a.arc if a.arc is not None else min(360.0, 0.72 * F)
    K = synth.default_K(W, H, f=525.0 * W / 640.0)
    t0 = time.time()
    frames, ext_gt, _ = synth.render_orbit_frames_torch(F, W, H, dev, arc_deg=arc, seed=7, K=K)
    # poses: ground truth + small noise (stands in for calibrate / solvePnP / adjustPose, out of scope)
    rng = np.random.default_rng(5)
    ext = ext_gt.copy()
    for f in range(F):
        ext[f, :, :3] = synth.rodrigues(rng.normal(0, 5e-4, 3)) @ ext_gt[f, :, :3]
        ext[f, :, 3] += rng.normal(0, 2e-3, 3)
    torch.cuda.synchronize()
    t_render = time.time() - t0
    pipe = ClipPipeline(H, W, N, batch=max(1, min(a.batch, -(-F // world) + 1)), device=dev, ctx=ctx)
    d = dist if use_dist else None

    def step(timers=None):
        if a.ba_window > 0:
            o = pipe.run(frames, K, ext, ba=False, dist=d, timers=timers)
            o["windows"] = pipe.adjust_windows(o, K, ext, window=a.ba_window, stride=a.ba_stride or max(1, a.ba_window // 2),
                                               ftol=1e-4, timers=timers, dist=d,
                                               order=a.ba_order or "wavefront",
                                               streams=a.ba_streams, batched=bool(a.ba_batched),
                                               max_nfev=a.ba_window_max_nfev or None)["windows"]
            return o
        return pipe.run(frames, K, ext, ba=not a.no_ba, ftol=1e-4, verbose=a.verbose, dist=d, timers=timers)

    for _ in range(a.warmup):
        out = step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # One extra, untimed step with events on EVERY launch gives the complete per-kernel table and names the dominant kernel
    # (largest share of device time).  Inside the timed steps only THAT kernel's launches are bracketed by events (level 3):
    # every bracketed launch costs the stream a bubble, and bracketing all launches of >= 64 workgroups (level 2, what
    # rounds 1-3 did; --profile-timed big) made a BA iteration 1.36 ms where it is 1.24 ms unobserved.
    prof_full = {}
    if not a.no_profile:
        ctx.profile(1)
        step()
        prof_full = ctx.profile_report()
        ctx.profile(0)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
    dominant_name = max(prof_full.items(), key=lambda kv: kv[1][1])[0] if prof_full else None
    if a.no_profile or dominant_name is None:
        ctx.profile(0)
    elif a.profile_timed == "big":
        ctx.profile(2)
    else:
        ctx.profile(3, only=dominant_name)
    timers = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(timers)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile(0)
    STAGES = ("detect", "match", "link", "triangulate", "ba", "ba_solve", "ba_windows")
    el = torch.tensor([elapsed] + [timers.get(k, 0.0) for k in STAGES],
                      dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el[0])
    stage_ms = {k: float(v) / a.steps for k, v in zip(STAGES, el[1:])}

    kp_count = out["kp_count"]
    pair_evals = float(np.sum(kp_count[:-1].astype(np.float64) * kp_count[1:].astype(np.float64)))
    res = out.get("ba")
    n_obs = out["n_obs"]
    nfev = res.nfev if res is not None else 0
    value = pair_evals / (stage_ms["match"] * 1e-3) if stage_ms["match"] > 0 else 0.0
    ba_rps = n_obs * nfev / (stage_ms["ba"] * 1e-3) if stage_ms["ba"] > 0 else 0.0

    # ---- roofline of the dominant kernel (this rank's launches; rank 0 reports) ----
    lv_w, lv_h, _, _ = ops.orb_level_sizes(H, W, pipe.prm)
    (p_lo, p_hi), (f_lo, f_hi) = __import__("meatmodeler_amd.parallel", fromlist=["x"]).pair_block(F, rank, world)
    kpl = kp_count[f_lo:f_hi].astype(np.float64)
    cfg = dict(frames=F, frames_local=f_hi - f_lo, nfeatures=N, levels=list(zip(lv_w.tolist(), lv_h.tolist())),
               pairs_local=p_hi - p_lo, pair_evals_local=float(np.sum(kpl[:-1] * kpl[1:])) if len(kpl) > 1 else 0.0,
               kp_total_local=float(kpl.sum()), n_obs_local=out.get("n_obs_local", n_obs),
               n_points_local=out["n_tracks"] // world, ba_iters=res.njev if res is not None else 0, ba_nfev=nfev,
               n_pairs=out.get("n_pairs", 0), cam_span=out.get("cam_span", F))
    def table(prof_, steps_):
        rows = []
        for name, (cnt, ms) in sorted(prof_.items(), key=lambda kv: -kv[1][1]):
            by, lops, fl = algorithmic_work(name, cfg)
            if name.startswith("chol_band_"):
                by, fl = (by or 0) * cnt / steps_, fl * cnt / steps_
            per = ms / steps_
            row = dict(kernel=name, launches_per_step=cnt / steps_, ms_per_step=per, avg_us=1e3 * ms / max(cnt, 1))
            if by:
                row["algorithmic_GBps"] = by / (per * 1e-3) / 1e9
            if lops:
                row["valu_Tlops"] = lops / (per * 1e-3) / 1e12
                if name == "schur_pairs_kernel":      # f64 lane-instructions: half the 32-bit issue rate
                    row["f64_valu_frac"] = row["valu_Tlops"] / (VALU_PEAK_TLOPS / 2)
                    row["note"] = ("bound by the latency of its dependent gathers at two waves per SIMD (236 VGPRs), not by arithmetic: "
                                   "ablation in csrc/schur.hip")
            if fl:
                row["mfma_f64_TFLOPs"] = fl / (per * 1e-3) / 1e12
            rows.append(row)
        return rows

    kernels_full = table(prof_full, 1)
    kernels = []
    for name, (cnt, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        by, lops, fl = algorithmic_work(name, cfg)
        if name.startswith("chol_band_"):
            by, fl = (by or 0) * cnt / a.steps, fl * cnt / a.steps
        per = ms / a.steps
        row = dict(kernel=name, launches_per_step=cnt / a.steps, ms_per_step=per, avg_us=1e3 * ms / max(cnt, 1))
        if by:
            row["algorithmic_GBps"] = by / (per * 1e-3) / 1e9
        if lops:
            row["valu_Tlops"] = lops / (per * 1e-3) / 1e12
            if name == "schur_pairs_kernel":
                row["f64_valu_frac"] = row["valu_Tlops"] / (VALU_PEAK_TLOPS / 2)
        if fl:
            row["mfma_f64_TFLOPs"] = fl / (per * 1e-3) / 1e12
        kernels.append(row)
    # Dominant kernel = largest share of device time in the fully profiled extra step; its roofline figures come from the
    # events around its launches inside the timed steps.  The other rows of "kernels" are the extra step's.
    timed = {k["kernel"]: k for k in kernels}
    for k in kernels:
        k["events_in_timed_region"] = True
    for k in kernels_full:
        if k["kernel"] not in timed:
            kernels.append(dict(k, events_in_timed_region=False))
    kernels.sort(key=lambda k: -k["ms_per_step"])
    dom_full = kernels_full[0] if kernels_full else (kernels[0] if kernels else None)
    roofline = None
    if dom_full is not None:
        dom = timed.get(dom_full["kernel"], dom_full)
        in_timed = dom_full["kernel"] in timed
        pmc = {}
        try:
            pmc_file = next(f for f in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["kernels"].get(dom["kernel"], {})
        except Exception:
            pmc_file = None
        traffic = None
        if F == 500 and N == 4000 and "FETCH_SIZE_KB_per_launch" in pmc and "WRITE_SIZE_KB_per_launch" in pmc:
            traffic = (pmc["FETCH_SIZE_KB_per_launch"] + pmc["WRITE_SIZE_KB_per_launch"]) * 1024.0
        if dom["kernel"] == "chol_band_fused_kernel":
            nblk_ = (6 * F + 63) // 64
            bwb_ = min((6 * cfg.get("cam_span", F) + 5 + 63) // 64, nblk_)
            note = ("single-launch banded Cholesky, eliminated from both ends of the band at once: a chain of %d dependent "
                    "64-column steps (factor 64x64 -> solve -> update; %d one-ended), bound by the latency of that chain, "
                    "not by MFMA throughput" % ((nblk_ - bwb_ + 1) // 2 + bwb_ if nblk_ - bwb_ >= 4 else nblk_, nblk_))
        else:
            note = None
        if "mfma_f64_TFLOPs" in dom:
            roofline = dict(kernel=dom["kernel"], bound="mfma", achieved=dom["mfma_f64_TFLOPs"], peak=MFMA_F64_PEAK_TF,
                            unit="TFLOP/s", frac=dom["mfma_f64_TFLOPs"] / MFMA_F64_PEAK_TF, traffic=traffic,
                            avg_launch_us=dom["avg_us"], events_in_timed_region=in_timed)
        else:
            ach = dom.get("algorithmic_GBps", 0.0)
            roofline = dict(kernel=dom["kernel"], bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBS, traffic=traffic, avg_launch_us=dom["avg_us"],
                            events_in_timed_region=in_timed)
            if "valu_Tlops" in dom:   # VALU-issue-bound kernels: the binding roof, stated beside the HBM one
                roofline["valu_achieved_Tlops"] = dom["valu_Tlops"]
                roofline["valu_peak_Tlops"] = VALU_PEAK_TLOPS
                roofline["valu_frac"] = dom["valu_Tlops"] / VALU_PEAK_TLOPS
        if note:
            roofline["note"] = note
        if traffic is not None:
            roofline["traffic_note"] = ("NOT measured by this run: FETCH_SIZE + WRITE_SIZE per launch read from the committed "
                                        "file profiles/%s (separate rocprofv3 --pmc passes of this same command, "
                                        "tools/gpu_evidence.sh); FETCH not doubled: 8-byte gathers are uncalibrated" % pmc_file)
    bf = next((k for k in kernels if k["kernel"].startswith("bf_knn2_")), None)
    if bf is not None and "valu_Tlops" in bf:
        bf = dict(bf)
        # the roof SURVEY section 8(d) fixes for the xor / popcount formulation (16 lane-ops per pair at the nominal
        # 78.6 T lane-ops/s = 4.9 T pairs/s) ...
        bf["T_pairs_per_s"] = bf["valu_Tlops"] / 16.0
        bf["frac_of_16op_valu_roof"] = bf["valu_Tlops"] / VALU_PEAK_TLOPS
        # ... which the kernel no longer lives under: the distances are computed on the matrix cores (descriptors as +1 / -1
        # FP4 values, v_mfma_scale_f32_32x32x64_f8f6f4: 512 flop per descriptor pair, exact), the vector unit keeps the
        # two-smallest bookkeeping (2 packed 16-bit instructions per pair)
        bf["mfma_fp4_PFLOPs"] = bf["T_pairs_per_s"] * 512.0 / 1e3
        bf["mfma_fp4_peak_PFLOPs"] = MFMA_FP4_PEAK_PF
        bf["frac_of_fp4_mfma_roof"] = bf["mfma_fp4_PFLOPs"] / MFMA_FP4_PEAK_PF
        # the 10 PF figure is the spec at 2.4 GHz on operands that do not toggle; on random +1 / -1 operands with every SIMD
        # busy the same instruction sustains 6.4 PFLOP/s on this chip (1.8 GHz; tools/dev/mfma_rate.hip,
        # profiles/r03_mfma_rate.txt), and the kernel itself runs at 1.5-1.7 GHz (tools/dev/bf_inkernel_clock.sh)
        bf["mfma_fp4_measured_roof_PFLOPs"] = MFMA_FP4_MEASURED_PF
        bf["frac_of_measured_fp4_roof"] = bf["mfma_fp4_PFLOPs"] / MFMA_FP4_MEASURED_PF
        bf["note"] = ("matrix cores (FP4 operands); the vector unit only keeps a minimum over each lane's 16 accumulators per "
                      "train tile (13 instructions per 16 descriptor pairs), the second smallest inside the best tile is "
                      "recomputed once per query by xor / popcount; MM_BF_VARIANT=300 selects the round-2 FP4 kernel (2 "
                      "instructions per pair), 200 the int8 MFMA form, 114 the xor / popcount kernel (2.0 T pairs/s = 91 % of the "
                      "VALU issue roof of its instruction mix, profiles/r02_bf_pmc.txt); all return identical results")
        for k_ in ("valu_Tlops",):
            bf.pop(k_, None)

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(frames, N, F, nfev)

    if rank == 0:
        line = {
            "metric": "descriptor-match pairs/sec + BA residuals/sec, 500×1080p frames, 1/2/4/8 GPU",
            "value": value, "unit": "descriptor pairs/s (match stage)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8/int32 (match), f64 (BA)",
            "data": "synthetic",
            "config": {"workload": f"{F} frames {W}x{H}, {N} ORB kpts/frame, BF Hamming 2-NN + ratio, track linking, "
                                   "2-view DLT, full BA (TRF, Schur, f64 MFMA Cholesky)" + (" [BA off]" if a.no_ba else ""),
                       "frames": F, "height": H, "width": W, "nfeatures": N, "parallelism": f"pairs/points x{world}"},
            "ba_residuals_per_s": ba_rps, "frames_per_s": F / (elapsed / a.steps), "stage_ms": stage_ms,
            # the trust-region path on this outlier-laden clip is chaotic (the iteration count moves with rounding-level
            # changes), so the per-iteration time is the figure to compare between builds, not ms_per_step
            "ba": None if res is None else {
                "nfev": nfev, "njev": res.njev, "iterations": getattr(res, "iterations", None),
                "ms_per_iteration": stage_ms["ba_solve"] / max(getattr(res, "iterations", 0) or 1, 1),
                "host_segments_ms_last_step": getattr(res, "host_segments_ms", None),
                # N > 1: the same library loop (mm_ba_trf_dist); its exchange points call back into torch.distributed
                "driver": "mm_ba_trf_dist" if world > 1 and getattr(res, "collectives", 0) else
                          ("mm_ba_trf" if getattr(res, "host_segments_ms", {}).get("library") is not None else "python"),
                "collectives": getattr(res, "collectives", None),
                "collectives_per_evaluation": (getattr(res, "collectives", 0) or 0) / max(nfev, 1),
                "chol_fallbacks": getattr(res, "chol_fallbacks", None)},
            "problem": {"keypoints": int(kp_count.sum()), "descriptor_pairs": pair_evals,
                        "matches": int(out["match_count"].sum()), "tracks": out["n_tracks"], "observations": n_obs,
                        "ba_nfev": nfev, "ba_status": res.status if res is not None else None,
                        "ba_cost": res.cost if res is not None else None, "cam_span": out.get("cam_span"),
                        "schur_pairs": out.get("n_pairs"), "render_s": t_render},
            "sliding_window_ba": None if "windows" not in out else {
                "window": a.ba_window, "stride": a.ba_stride or max(1, a.ba_window // 2), "windows": len(out["windows"]),
                "order": a.ba_order or "wavefront", "streams": a.ba_streams, "max_nfev_per_window": a.ba_window_max_nfev or None,
                "batched": bool(a.ba_batched) and (a.ba_order or "wavefront") == "wavefront",
                "ms": stage_ms["ba_windows"], "nfev_total": int(sum(w["nfev"] for w in out["windows"])),
                "observations_total": int(sum(w["observations"] for w in out["windows"])),
                "residual_evals_per_s": sum(w["observations"] * w["nfev"] for w in out["windows"]) /
                                        max(stage_ms["ba_windows"] * 1e-3, 1e-9),
                "status_counts": {str(k): int(sum(1 for w in out["windows"] if w["status"] == k))
                                  for k in sorted({w["status"] for w in out["windows"]})},
                "per_window": [[w["lo"], w["hi"], w["points"], w["observations"], w["nfev"], w["status"],
                                float(f"{w['cost']:.6g}")] for w in out["windows"]],
                "per_window_columns": ["lo", "hi", "points", "observations", "nfev", "status", "cost"]},
            "roofline": roofline,
            "bf_knn2": bf,
            "kernels": kernels[:12],
            "kernels_all_launches_extra_step": kernels_full[:16],
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(frames, nfeatures, n_frames, gpu_nfev):
    """The CPU oracle ("port" of the reference's CPU path: the C restatement of ORB + BF matching, and the reference's
    own SciPy TRF/LSMR recipe on the NumPy restatement of pointFun) timed on this box's host cores on a bounded sample
    of the same workload: one frame / one pair on one core, then one frame / pair per available core in parallel
    (threads around the C calls, which release the GIL) for the all-core rate; BA on 200 frames / 60 000 points / 480 000
    observations (a third of the GPU's problem; SciPy's cost per residual evaluation is flat in that range).
    `gpu_nfev`: evaluations the GPU solve took on the clip -- the end-to-end estimate charges the CPU path the same number."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orb_oracle as oo
    from oracle import ba_oracle as bo
    from meatmodeler_amd import synth
    from meatmodeler_amd.orb_pattern import brief_pattern
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, frames.shape[0]))      # every core this process may run on (the count is reported)
    host = frames[:cores].cpu().numpy()
    pat = brief_pattern()
    oo.detect_compute(host[0][:64, :64].copy(), 50, pat)           # (builds / loads the library outside the timings)
    t0 = time.perf_counter()
    d0 = oo.detect_compute(host[0], nfeatures, pat)
    t_orb1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        dets = list(ex.map(lambda im: oo.detect_compute(im, nfeatures, pat), host))
    t_orb_all = (time.perf_counter() - t0) / len(host)             # wall time per frame with all cores busy
    a_, b_ = dets[0]["desc"], dets[-1]["desc"]
    t0 = time.perf_counter()
    idx, dist_ = oo.bf_knn2(a_, b_)
    oo.ratio_filter(idx, dist_, 0.75)
    t_match1 = time.perf_counter() - t0
    pairs = float(len(a_)) * float(len(b_))

    def one_pair(k):
        i_, d_ = oo.bf_knn2(dets[k]["desc"], dets[(k + 1) % len(dets)]["desc"])
        oo.ratio_filter(i_, d_, 0.75)
        return float(len(dets[k]["desc"])) * float(len(dets[(k + 1) % len(dets)]["desc"]))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        pairs_all = sum(ex.map(one_pair, range(len(dets))))
    t_match_all = time.perf_counter() - t0
    Fb, Pb = 200, 60000
    pr = synth.make_ba_problem(Fb, Pb, 8, seed=1)
    t0 = time.perf_counter()
    _, _, r = bo.adjust_points(pr["ext"], pr["K"], pr["pts0"][:, None, :], pr["obs"], pr["fi"], pr["pi"],
                               return_result=True)
    t_ba = time.perf_counter() - t0
    O = len(pr["fi"])
    ba_rps = O * r.nfev / t_ba
    obs_per_frame = 0.9 * nfeatures
    frame_s = t_orb_all + t_match_all / max(len(dets), 1) + obs_per_frame * max(gpu_nfev, 1) / ba_rps
    return {"value": pairs_all / t_match_all, "unit": "descriptor pairs/s", "cores": cores, "kind": "port",
            "sample": f"ORB: 1 frame on 1 core {t_orb1 * 1e3:.0f} ms, {len(host)} frames on {cores} cores "
                      f"{t_orb_all * 1e3:.0f} ms per frame; BF match {len(a_)}x{len(b_)}: 1 pair on 1 core "
                      f"{t_match1 * 1e3:.0f} ms, {len(dets)} pairs on {cores} cores {t_match_all * 1e3:.0f} ms in all; "
                      f"SciPy TRF+LSMR BA (the reference's recipe, SciPy's own threading) on {Fb} frames / {Pb} points / "
                      f"{O} observations: {t_ba:.2f} s, {r.nfev} nfev",
            "value_1core": pairs / t_match1, "orb_ms_per_frame_1core": t_orb1 * 1e3,
            "orb_ms_per_frame_all_cores": t_orb_all * 1e3, "match_ms_per_pair_1core": t_match1 * 1e3,
            "ba_residuals_per_s": ba_rps, "ba_nfev_cpu_sample": int(r.nfev), "ba_nfev_charged": int(max(gpu_nfev, 1)),
            "frames_per_s_estimate": 1.0 / frame_s,
            "frames_per_s_estimate_note": "ORB + matching at the all-core rates, BA at the sample's residual rate for as many "
                                          "evaluations as the GPU solve took on this clip",
            "host_cpu_count": os.cpu_count(),
            "note": "the ORB / matching legs are an UNOPTIMISED scalar C restatement of the algorithm (oracle/orb_oracle.c, "
                    "no SIMD; OpenCV's own ORB is more than an order of magnitude faster per core and is not installable "
                    "here), run on every core of this process's affinity mask; a stated baseline, not a tuned competitor"}


if __name__ == "__main__":
    main()
