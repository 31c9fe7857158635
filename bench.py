#!/usr/bin/env python3
"""Benchmark of the reconstruction hot path on MI355X (contract: see the task statement).

One "step" = one pass of the whole hot path over this rank's shard of synthetic frames, inputs
already resident in HBM:  cloud_big reset -> A6 for every frame (fused reproject + SE(3), per-frame
voxel grid), appended in frame order -> [N>1: index-slice partition + one RCCL all-to-all of the
per-frame voxel clouds] -> combined 2.5-D merge (pose.cpp:530) [N>1: all-gather of the merged slices].  Workload = BASELINE.json configs[1]: synthetic 1280x720 dense
(jump_pixels 1), 200 frames per GPU, voxel_size 0.05.

After the timed region the default run also reports, outside `value`: the PCIe-inclusive rate (host inputs), the rate
with the reference's statistical outlier removal on (`sor_on_frames_per_sec`, checked against the oracle), the bit-for-bit
verification of the whole step against the oracle, the distance of the merged cloud to the reference's own std::sort
summation order (`verification.vs_reference_sort_order*`), and the oracle timed as `cpu_baseline`.

    python bench.py                       # N=1, defaults finish in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_TRAFFIC_FILE = "r04_pmc_traffic.json"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; what a plain device copy reaches on the box is measured below
                       # (roofline.measured_copy_GBps: 4.7-5.6 TB/s read+write over the boxes seen so far)


def _gen(args):
    from online_3d_reconstruction_amd import synth
    i, rows, cols, invalid = args
    return synth.make_frame(i, rows, cols, invalid_frac=invalid)


def generate_frames(start, count, rows, cols, invalid, workers):
    """host-side synthetic frames (before the GPU is touched: the pool forks)"""
    disp = np.empty((count, rows, cols), np.uint8)
    bgr = np.empty((count, rows, cols, 3), np.uint8)
    jobs = [(start + i, rows, cols, invalid) for i in range(count)]
    if workers > 1:
        with ProcessPoolExecutor(workers) as ex:
            for i, (d, c) in enumerate(ex.map(_gen, jobs, chunksize=4)):
                disp[i], bgr[i] = d, c
    else:
        for i, j in enumerate(jobs):
            disp[i], bgr[i] = _gen(j)
    return disp, bgr


def _grid_shape(rows, cols, jump, bb=20, cutout=8):
    """(Ny, Nx) of the grid pass, pose_functions.cpp:1094-1096,638"""
    cs = int(cols / cutout)
    if jump <= 0:
        return 0, 0
    return max(0, -(-(rows - 2 * bb) // jump)), max(0, -(-(cols - bb - cs) // jump))


def baseline_config(args, total_frames):
    """which entry of BASELINE.json `configs` (0-based, as in the file) this run's shape is"""
    shape = (args.rows, args.cols, args.jump_pixels, args.voxel_size, args.min_points)
    if shape == (720, 1280, 1, 0.05, 1):
        return "BASELINE.json configs[2]: 2000 frames in all" if total_frames == 2000 else "BASELINE.json configs[1]"
    if shape == (1080, 1920, 1, 0.02, 3):
        return "BASELINE.json configs[3]"
    if shape == (2160, 4096, 4, 0.05, 1):
        return "BASELINE.json configs[4]" + (": 2000 frames in all" if total_frames == 2000 else " shape")
    if shape == (720, 1280, 15, 0.05, 1):
        return "BASELINE.json configs[0] shape, synthetic frames"
    return "not a BASELINE.json config"


def kernel_sources_sha1():
    """hash of the device sources: ties a PMC traffic file to the kernels it was measured on"""
    import glob
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "online_3d_reconstruction_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")) +
                       glob.glob(os.path.join(base, "kernels", "*.inc"))):
        h.update(open(path, "rb").read())
    return h.hexdigest()


def cpu_baseline(disp, bgr, poses, Q, voxel_size, jump, threads, n_frames, sor=False):
    """The CPU oracle (a port of the reference arithmetic, oracle/) timed on this host: A6 for
    n_frames frames on `threads` frame-parallel POSIX threads — the reference's own fan-out
    (pose.cpp:392-413) — appended in frame order, then the combined merge (pose.cpp:530)."""
    from oracle import orc
    orc.lib()
    t0 = time.perf_counter()
    orc.run_frames(disp[:n_frames], bgr[:n_frames], Q, poses[:n_frames], voxel_size, jump_pixels=jump, sor=sor,
                   threads=threads, want_clouds=False)
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def compare_clouds_by_cell(got, ref, voxel_size):
    """Two merged clouds (one point per XY cell, ascending cell order) -> how far apart they are.  Cells are matched by
    their (ix, iy) index so that the comparison still means something if the two differ in occupancy."""
    vs = np.float32(voxel_size)

    def cell_ids(c):
        return (np.floor(c["y"] / vs).astype(np.int64) << 32) + (np.floor(c["x"] / vs).astype(np.int64) & 0xFFFFFFFF)

    ka, kb = cell_ids(got), cell_ids(ref)
    same_cells = len(ka) == len(kb) and bool(np.array_equal(ka, kb))
    if same_cells:
        ia = ib = slice(None)
        common = len(ka)
    else:
        _, ia, ib = np.intersect1d(ka, kb, assume_unique=False, return_indices=True)
        common = len(ia)
    a, b = got[ia], ref[ib]
    out = {"count_equal": len(got) == len(ref), "cells_equal": same_cells, "cells_compared": int(common),
           "rgba_equal": bool(np.array_equal(a["rgba"], b["rgba"])),
           "rgba_differing_cells": int((a["rgba"] != b["rgba"]).sum())}
    for ax in "xyz":
        d = np.abs(a[ax].astype(np.float64) - b[ax].astype(np.float64))
        out["max_abs_d" + ax] = float(d.max()) if common else 0.0
        out["mean_abs_d" + ax] = float(d.mean()) if common else 0.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=200, help="frames per GPU (weak scaling)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="total frames over all GPUs (strong scaling; BASELINE configs[2]: --total-frames 2000 on 8 GPUs); "
                         "overrides --frames with total/N per GPU")
    ap.add_argument("--rows", type=int, default=720)
    ap.add_argument("--cols", type=int, default=1280)
    ap.add_argument("--jump-pixels", type=int, default=1)
    ap.add_argument("--voxel-size", type=float, default=0.05)
    ap.add_argument("--min-points", type=int, default=1)
    ap.add_argument("--invalid-frac", type=float, default=0.0)
    ap.add_argument("--sor", action="store_true", help="statistical outlier removal on (the reference's full per-frame "
                                                      "path, pose_functions.cpp:1673-1686); off in the headline config")
    ap.add_argument("--blur-kernel", type=int, default=1, help="> 1: bilateral filter on every disparity image first "
                    "(--blur_kernel of the reference, README.md:50 uses 30); informational, implies --no-cpu-baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (cpu_baseline AND the bit-for-bit "
                    "verification of the GPU clouds against the oracle, which reuses it)")
    ap.add_argument("--no-reference-order", action="store_true", help="skip the comparison with the oracle run in the "
                    "reference's own (libstdc++ std::sort) summation order")
    ap.add_argument("--no-sor-leg", action="store_true", help="skip the extra leg with statistical outlier removal on")
    ap.add_argument("--sor-cpu-frames", type=int, default=7, help="frames of the CPU sample with outlier removal on")
    ap.add_argument("--no-pcie-step", action="store_true", help="skip the extra untimed step with HOST inputs "
                    "(pcie_inclusive_frames_per_sec)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="hand the library HOST buffers (frames cross PCIe inside the timed region); reported as "
                         "metric frames_per_sec_pcie_inclusive, never the headline value")
    ap.add_argument("--pinned", action="store_true", help="with --host-inputs: page-locked host buffers")
    ap.add_argument("--verify-max-frames", type=int, default=2000, help="N > 1: rank 0 re-creates the frames of all ranks and "
                    "checks the merged cloud against the oracle when there are at most this many in all")
    ap.add_argument("--cpu-frames", type=int, default=200)
    ap.add_argument("--cpu-threads", type=int, default=7)
    ap.add_argument("--gen-workers", type=int, default=min(16, os.cpu_count() or 1))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes through torch's launcher
            # (before anything here touches the GPU; never an exec) and relay rank 0's JSON line, which the children
            # print on the stdout they inherit
            import socket
            import subprocess
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
        args.gpus = world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # ---- inputs (host) -----------------------------------------------------------------------------
    from online_3d_reconstruction_amd import synth
    scaling = "weak"
    if args.total_frames > 0:
        if args.total_frames % world:
            sys.exit("--total-frames must be a multiple of the number of GPUs")
        args.frames = args.total_frames // world
        scaling = "strong"
    F = args.frames
    first = rank * F  # contiguous block of frames per rank
    n_cand_host = (lambda g: g[0] * g[1])(_grid_shape(args.rows, args.cols, args.jump_pixels))

    def host_can_verify(total_frames):
        # the oracle holds all ranks' cloud_big twice and its sort records: ~3.5 x 16 bytes per candidate point
        try:
            import psutil
            need = 3.5 * 16 * n_cand_host * total_frames + 4.0 * args.rows * args.cols * total_frames
            return psutil.virtual_memory().available > 1.25 * need
        except Exception:
            return total_frames <= 400

    # N > 1: rank 0 checks the merged cloud against the oracle's run over the frames of ALL ranks; they are generated
    # here, before anything touches the GPU (the generator forks a process pool)
    all_disp = all_bgr = None
    verify_all = (rank == 0 and world > 1 and not args.no_cpu_baseline and args.blur_kernel <= 1 and
                  F * world <= args.verify_max_frames and host_can_verify(F * world))
    if verify_all:
        all_disp, all_bgr = generate_frames(0, F * world, args.rows, args.cols, args.invalid_frac, args.gen_workers)
        disp_h, bgr_h = all_disp[:F], all_bgr[:F]
    else:
        disp_h, bgr_h = generate_frames(first, F, args.rows, args.cols, args.invalid_frac, args.gen_workers)
    poses_h = synth.make_poses(first, F)
    Q = synth.camera_Q(args.rows, args.cols)

    import torch
    import torch.distributed as dist

    import online_3d_reconstruction_amd as o3dr
    from online_3d_reconstruction_amd import _lib as L
    from online_3d_reconstruction_amd import dist as o3dist

    # rehearsal on a one-GPU box: O3DR_BENCH_REHEARSAL=1 puts every rank on cuda:0 and runs the
    # collectives over gloo on host copies (never used for reported numbers)
    rehearsal = os.environ.get("O3DR_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = torch.device("cpu") if rehearsal else None
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # N > 1: the library and torch's collectives work on ONE explicit stream, so that the exchange needs no host waits to
    # order them (the NULL stream handle would mean "the context's own stream" to o3dr_ctx_set_stream)
    stream = torch.cuda.Stream(device=dev) if world > 1 else torch.cuda.current_stream()
    if world > 1:
        torch.cuda.set_stream(stream)

    ctx = o3dr.Context(local_rank, Q=Q, params=o3dr.Params(jump_pixels=args.jump_pixels, voxel_size=args.voxel_size,
                                                          min_points_per_voxel=args.min_points, sor_enable=args.sor,
                                                          blur_kernel=args.blur_kernel), stream=stream)
    if args.host_inputs and args.pinned:
        disp = torch.from_numpy(disp_h).pin_memory()
        bgr = torch.from_numpy(bgr_h).pin_memory()
        poses = torch.from_numpy(poses_h).pin_memory()
    elif args.host_inputs:
        disp, bgr, poses = disp_h, bgr_h, poses_h
    else:
        disp = torch.from_numpy(disp_h).to(dev)
        bgr = torch.from_numpy(bgr_h).to(dev)
        poses = torch.from_numpy(poses_h).to(dev)
    n_cand = ctx.max_points(args.rows, args.cols)
    ctx.cloudBigReserve(F * n_cand if world == 1 else F * n_cand)

    # valid pixels per frame (for the algorithmic byte counts), from the host copy
    bb, cs = 20, args.cols // 8
    roi = disp_h[:, bb:args.rows - bb:max(args.jump_pixels, 1), cs:args.cols - bb:max(args.jump_pixels, 1)]
    n_valid_total = int((roi > 64).sum())

    state = {}

    def step():
        ctx.cloudBigReset()
        ctx.accumulateFrames(disp, bgr, poses)
        if world > 1:
            # index-slice partition + one all-to-all + local merge + all-gather of the merged slices
            out, m1 = o3dist.merge_partitioned(ctx, dev, comm_device=comm_dev)
        else:
            m1, _ = ctx.cloudBigSize()
            out = ctx.finalize(device=dev)
        if world > 1:
            state["exchange"] = dict(o3dist.last_stats)
        state["m1_total"] = m1
        state["m2"] = int(out.shape[0])
        state["out"] = out
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()

    # one untimed, fully bracketed step: which kernel dominates?
    barrier()
    ctx.profileReset()
    ctx.profileEnable(-1, True)
    step()
    per_kernel = {L.KERNEL_NAMES[k]: ctx.profileRead(k) for k in range(len(L.KERNEL_NAMES))}
    ctx.profileEnable(-1, False)
    dom = max((k for k in range(len(L.KERNEL_NAMES)) if k != L.K_OTHER), key=lambda k: per_kernel[L.KERNEL_NAMES[k]][0])

    # ---- timed region: exactly K steps, only the dominant kernel bracketed by HIP events -----------
    ctx.profileReset()
    ctx.profileEnable(dom, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_launches = ctx.profileRead(dom)
    stats = ctx.profileStatsAll()
    ctx.profileEnable(dom, False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev or dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    frames_total = F * world * args.steps
    fps = frames_total / dt
    m1_total, m2 = state["m1_total"], state["m2"]

    # ---- algorithmic bytes per step (DESIGN.md "Kernels and rooflines") --------------------------------
    rec_passes, vox_in, vox_out, _unused, sort_recs = (v / args.steps for v in stats[:5])  # device counters, per step
    nv = n_valid_total  # valid points of this rank's frames (per step)
    merge_n = m1_total // world  # points entering the combined merge on this rank (its index slice)
    m1 = m1_total // world       # per-frame voxels of this rank's frames
    merge_in = vox_in - nv       # points that entered whole-cloud grids (the merge), not per-frame ones
    bytes_per_step = {
        "reproject_count": 1 * n_cand * F,                     # 1 B disparity per candidate
        "reproject_emit": 4 * n_cand * F + 16 * nv,            # 1 B disparity + 3 B colour in, 16 B point out
        "voxel_keys": 0.25 * merge_in,                         # batch path: indices come from the emit pass; the merge reads its run-head flags
        "radix_hist": 4 * rec_passes,                          # 4 B index per record per pass
        "radix_scatter": 16 * rec_passes - 4 * sort_recs,      # (index,id) in and out; pass 0 has no id to read
        "run_segments": 4.5 * sort_recs + 4 * vox_out,         # index read once, head flags out and in, run starts written
        "centroid": 20 * nv + 16 * m1,                         # id + gathered point in, centroid out
        "centroid_runs": 16 * merge_n + 16 * m2,               # merge: points in (group runs are contiguous), cells out
    }
    dom_name = L.KERNEL_NAMES[dom]
    achieved = bytes_per_step[dom_name] * args.steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "traffic_from_profile": None, "traffic_measured_in_this_run": False,
                "avg_launch_us": round(dom_ms * 1e3 / max(dom_launches, 1), 2), "launches": int(dom_launches),
                "algorithmic_bytes_per_launch": int(bytes_per_step[dom_name] * args.steps / max(dom_launches, 1))}
    # what a plain device-to-device copy reaches on this box (read + write bytes): context for `peak`
    try:
        src_t = torch.empty(1 << 28, dtype=torch.int32, device=dev)  # 1 GiB
        dst_t = torch.empty_like(src_t)
        for _ in range(2):
            dst_t.copy_(src_t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst_t.copy_(src_t)
        e1.record()
        torch.cuda.synchronize()
        roofline["measured_copy_GBps"] = round(5 * 2 * src_t.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src_t, dst_t
    except Exception:
        roofline["measured_copy_GBps"] = None
    # HBM bytes per launch of the dominant kernel from the PMC passes kept under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
    # MI355X_MICROARCH.md "HBM"); only reported when that file was measured on the same kernel sources as this run
    # (a stored, builder-side measurement: `traffic_from_profile` says so; this run itself collects no PMC counters)
    traffic_file = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
    roofline["traffic_source"] = None
    if os.path.exists(traffic_file):
        try:
            tj = json.load(open(traffic_file))
            if tj.get("kernel") == dom_name and tj.get("kernel_sources_sha1") == kernel_sources_sha1():
                roofline["traffic"] = roofline["traffic_from_profile"] = tj.get("hbm_bytes_per_launch")
                roofline["traffic_source"] = ("profiles/" + PMC_TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                              "bench command on the same device sources, collected by the builder: not measured in this run)")
        except Exception:
            pass
    # SURVEY 8d end-to-end figure: B_frame = 4N + 16Nv + 16Nv + 16M1 per frame, B_final = 16 SUM(M1) + 16 M2
    b_frame = (4 * n_cand * F + 32 * nv + 16 * (m1_total // world)) / F
    b_final = 16 * merge_n + 16 * m2
    e2e_gbs = (b_frame * F + b_final) * args.steps * world / dt / 1e9

    result = {
        "metric": "frames_per_sec_pcie_inclusive" if args.host_inputs else "frames_per_sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: gloo, one GPU)" if rehearsal else ""),
        "config": {"workload": f"synthetic {args.cols}x{args.rows} dense stereo, jump_pixels {args.jump_pixels}, "
                               f"{F} frames/GPU, voxel_size {args.voxel_size}, min_points_per_voxel {args.min_points}, "
                               f"SOR {'on' if args.sor else 'off'}, " + (f"blur_kernel {args.blur_kernel}, " if args.blur_kernel > 1 else "") + f"frames resident in HBM ({baseline_config(args, F * world)})",
                   "frames_per_gpu": F, "rows": args.rows, "cols": args.cols, "jump_pixels": args.jump_pixels,
                   "voxel_size": args.voxel_size, "parallelism": f"frame-sharded x{world}"},
        "mpoints_per_sec_into_global_cloud": round(m1_total * args.steps / dt / 1e6, 2),
        "points": {"candidates_per_frame": n_cand, "valid_per_step_per_gpu": nv, "per_frame_voxels_total": m1_total,
                   "merged_cells": m2},
        "roofline": roofline,
        "end_to_end": {"algorithmic_GBps": round(e2e_gbs, 2), "frac_of_hbm_peak": round(e2e_gbs / HBM_PEAK_GBS / world, 5),
                       "bytes_per_frame": int(b_frame), "bytes_final_merge": int(b_final)},
        "kernel_ms_per_step": {k: round(v[0], 3) for k, v in per_kernel.items()},
        "sort": {"records_per_step": int(sort_recs), "record_passes_per_step": int(rec_passes)},
    }
    if world > 1:  # collectives per step, host waits before the final gather, and what the exchange moved (dist.merge_partitioned)
        result["exchange"] = state.get("exchange")
        ex = state.get("exchange") or {}
        mine3 = torch.tensor([ex.get("points_local", 0), ex.get("points_sent_off_rank", 0), ex.get("points_received_off_rank", 0)],
                             dtype=torch.int64, device=comm_dev or dev)
        all3 = torch.empty(3 * world, dtype=torch.int64, device=comm_dev or dev)
        dist.all_gather_into_tensor(all3, mine3)
        all3 = all3.cpu().numpy().reshape(world, 3)
        sent_total = int(all3[:, 1].sum())
        # xGMI model (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, one link per peer pair): the all-to-all is bound by
        # the busiest rank's bytes over the links it uses; the figure is a MODEL, nothing here measures a link
        busiest = int(max(all3[:, 1].max(), all3[:, 2].max())) * 16
        result["exchange_volume"] = {
            "points_local_per_rank": [int(v) for v in all3[:, 0]], "points_sent_off_rank_per_rank": [int(v) for v in all3[:, 1]],
            "points_received_off_rank_per_rank": [int(v) for v in all3[:, 2]],
            "fraction_of_points_leaving_their_rank": round(sent_total / max(int(all3[:, 0].sum()), 1), 4),
            "bytes_over_links_total": sent_total * 16, "busiest_rank_bytes": busiest,
            "modelled_xgmi_ms_one_link": round(busiest / 153e9 * 1e3, 3),
            "modelled_xgmi_ms_links_to_all_peers": round(busiest / (153e9 * max(min(world - 1, 7), 1)) * 1e3, 3)}

    # ---- PCIe-inclusive rate: one extra, untimed-in-`value` step with HOST (pageable numpy) inputs -----------------
    if world == 1 and not args.host_inputs and not args.no_pcie_step:
        for k in range(2):  # (the first call allocates the context's staging buffers; the second one is timed)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.cloudBigReset()
            ctx.accumulateFrames(disp_h, bgr_h, poses_h)
            ctx.finalize(device=dev)
            torch.cuda.synchronize()
            result["pcie_inclusive_frames_per_sec"] = round(F / (time.perf_counter() - t0), 2)

    # ---- the reference's literal per-frame path: statistical outlier removal ON (pose_functions.cpp:1673-1686 runs it in
    # every per-frame call when jump_pixels > 0).  Reported next to the headline, which is measured with it off.
    sor_leg = (world == 1 and not args.sor and not args.no_sor_leg and not args.host_inputs and args.jump_pixels > 0 and
               args.blur_kernel <= 1)
    if sor_leg:
        prm_on = o3dr.Params(jump_pixels=args.jump_pixels, voxel_size=args.voxel_size, min_points_per_voxel=args.min_points,
                             sor_enable=True, blur_kernel=args.blur_kernel)
        ctx.set_params(prm_on)
        for k in range(2):  # (the first pass allocates the outlier removal's workspaces)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.cloudBigReset()
            ctx.accumulateFrames(disp, bgr, poses)
            n_sor = int(ctx.finalize(device=dev).shape[0])
            torch.cuda.synchronize()
            result["sor_on_frames_per_sec"] = round(F / (time.perf_counter() - t0), 2)
        result["sor_on"] = {"frames": F, "merged_cells": n_sor,
                            "what": "same frames, one untimed-in-`value` step with sor_enable=1 (mean_k 50, 1 sigma), inputs in HBM"}
        if rank == 0 and not args.no_cpu_baseline:
            from oracle import orc
            # (a) the CPU sample that is TIMED: a few frames on the reference's 7 threads
            ns = max(1, min(args.sor_cpu_frames, F))
            t0 = time.perf_counter()
            orc.run_frames(disp_h[:ns], bgr_h[:ns], Q, poses_h[:ns], args.voxel_size, jump_pixels=args.jump_pixels,
                           min_points_per_voxel=args.min_points, sor=True, threads=args.cpu_threads, want_clouds=False)
            t_sor = time.perf_counter() - t0
            # (b) the verification: ALL frames of the step with the outlier removal on, on as many host threads as there are
            # (the oracle's kd-tree-free exact search takes ~2.6 s per dense frame and core), against the GPU's clouds
            nv = F if (os.cpu_count() or 1) >= 32 else ns
            vthreads = max(args.cpu_threads, min(os.cpu_count() or 1, 128, nv))
            t0 = time.perf_counter()
            rbig, rsmall = orc.run_frames(disp_h[:nv], bgr_h[:nv], Q, poses_h[:nv], args.voxel_size, jump_pixels=args.jump_pixels,
                                          min_points_per_voxel=args.min_points, sor=True, threads=vthreads)
            t_ver = time.perf_counter() - t0
            ctx.cloudBigReset()
            ctx.accumulateFrames(disp[:nv], bgr[:nv], poses[:nv])
            gbig = ctx.cloudBigRead()
            gsmall = o3dr.api.points_from_torch(ctx.finalize(device=dev))
            result["sor_on"].update({
                "verified": bool(len(gbig) == len(rbig) and np.array_equal(gbig.view(np.uint32), rbig.view(np.uint32)) and
                                 len(gsmall) == len(rsmall) and np.array_equal(gsmall.view(np.uint32), rsmall.view(np.uint32))),
                "verified_frames": nv, "verification_threads": vthreads, "verification_seconds": round(t_ver, 1),
                "cpu_frames_per_sec": round(ns / t_sor, 3), "cpu_threads": args.cpu_threads,
                "cpu_seconds": round(t_sor, 1), "cpu_sample_frames": ns})
            del rbig, gbig
        ctx.set_params(o3dr.Params(jump_pixels=args.jump_pixels, voxel_size=args.voxel_size, min_points_per_voxel=args.min_points,
                                   sor_enable=args.sor, blur_kernel=args.blur_kernel))

    # ---- verification: the whole step of the headline config against the oracle, bit for bit -----------------------
    result["verified"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.blur_kernel <= 1:
        from oracle import orc
        ctx.cloudBigReset()
        ctx.accumulateFrames(disp, bgr, poses)
        big = ctx.cloudBigRead()
        small = o3dr.api.points_from_torch(ctx.finalize(device=dev))
        t0 = time.perf_counter()
        rbig, rsmall = orc.run_frames(disp_h, bgr_h, Q, poses_h, args.voxel_size, jump_pixels=args.jump_pixels,
                                      min_points_per_voxel=args.min_points, sor=args.sor, threads=args.cpu_threads)
        t_or = time.perf_counter() - t0
        ok_big = len(big) == len(rbig) and np.array_equal(big.view(np.uint32), rbig.view(np.uint32))
        ok_small = len(small) == len(rsmall) and np.array_equal(small.view(np.uint32), rsmall.view(np.uint32))
        result["verified"] = bool(ok_big and ok_small)
        result["verification"] = {"against": "oracle/o3dr_oracle.c (orc_run_frames) on the identical frames and poses",
                                  "cloud_big_points": int(len(big)), "oracle_cloud_big_points": int(len(rbig)),
                                  "cloud_big_bit_equal": bool(ok_big), "merged_points": int(len(small)),
                                  "oracle_merged_points": int(len(rsmall)), "merged_bit_equal": bool(ok_small),
                                  "oracle_seconds": round(t_or, 2)}
        del big
        # The reference's REAL summation order: PCL sorts its (index, point) records with std::sort, which is not stable
        # (pose_functions.cpp:1700), after z += 500 in fp32 (:1664-1666).  No fixture of the reference pins that order;
        # the oracle reproduces the libstdc++ call (oracle/o3dr_oracle_stdsort.cpp).  Two comparisons of the GPU's merged
        # cloud: (a) the same cloud_big merged in std::sort order, (b) the whole pipeline (per-frame grids too) in it.
        if not args.no_reference_order:
            t0 = time.perf_counter()
            std_small, _ = orc.downsample_pt_cloud(rbig, args.voxel_size, True, args.min_points, orc.ORDER_STDSORT)
            cmp_a = compare_clouds_by_cell(small, std_small, args.voxel_size)
            del std_small
            _, std_all = orc.run_frames(disp_h, bgr_h, Q, poses_h, args.voxel_size, jump_pixels=args.jump_pixels,
                                        min_points_per_voxel=args.min_points, sor=args.sor, threads=args.cpu_threads,
                                        order=orc.ORDER_STDSORT)
            cmp_b = compare_clouds_by_cell(small, std_all, args.voxel_size)
            del std_all
            cmp_a["points_per_cell"] = cmp_b["points_per_cell"] = round(len(rbig) / max(len(small), 1), 1)
            cmp_a["what"] = "GPU merged cloud vs the oracle's merge of the SAME cloud_big in libstdc++ std::sort order"
            cmp_b["what"] = ("GPU merged cloud vs the oracle's whole run (per-frame grids and merge) in libstdc++ std::sort "
                             "order: the reference's own arithmetic as far as it can be restated here")
            result["verification"]["vs_reference_sort_order"] = cmp_a
            result["verification"]["vs_reference_sort_order_whole_pipeline"] = cmp_b
            result["verification"]["reference_order_seconds"] = round(time.perf_counter() - t0, 2)
        del rbig

    # ---- N > 1: every rank's shard of cloud_big (hashed) and the merged cloud every rank holds after the exchange against
    # the oracle's run over the frames of ALL ranks
    if world > 1 and not args.no_cpu_baseline and args.blur_kernel <= 1:
        import hashlib
        ctx.cloudBigReset()
        ctx.accumulateFrames(disp, bgr, poses)
        shard = ctx.cloudBigRead()  # this rank's frames' per-frame voxels, before any exchange
        mine = np.zeros(28, np.uint8)
        mine[:8] = np.frombuffer(np.int64(len(shard)).tobytes(), np.uint8)
        mine[8:] = np.frombuffer(hashlib.sha1(shard.tobytes()).digest(), np.uint8)
        cd = comm_dev or dev
        allh = torch.empty(world * 28, dtype=torch.uint8, device=cd)
        dist.all_gather_into_tensor(allh, torch.from_numpy(mine).to(cd))
        allh = allh.cpu().numpy().reshape(world, 28)
        del shard
    if rank == 0 and world > 1 and not args.no_cpu_baseline and args.blur_kernel <= 1 and not verify_all:
        result["verification"] = {"skipped": f"{F * world} frames in all: above --verify-max-frames or the host's free memory"}
    elif rank == 0 and world > 1 and not args.no_cpu_baseline and args.blur_kernel <= 1:
        from oracle import orc
        all_poses = synth.make_poses(0, F * world)
        t0 = time.perf_counter()
        rbig, rsmall = orc.run_frames(all_disp, all_bgr, Q, all_poses, args.voxel_size, jump_pixels=args.jump_pixels,
                                      min_points_per_voxel=args.min_points, sor=args.sor, threads=args.cpu_threads)
        t_or = time.perf_counter() - t0
        small = o3dr.api.points_from_torch(state["out"])
        counts = [int(np.frombuffer(allh[r, :8].tobytes(), np.int64)[0]) for r in range(world)]
        ok_big = sum(counts) == len(rbig) == int(m1_total)
        shard_ok = []
        off = 0
        for r in range(world):  # the oracle's cloud_big is in frame order = rank order
            seg = rbig[off: off + counts[r]] if ok_big else rbig[:0]
            shard_ok.append(bool(ok_big and hashlib.sha1(seg.tobytes()).digest() == allh[r, 8:].tobytes()))
            off += counts[r]
        ok_small = len(small) == len(rsmall) and np.array_equal(small.view(np.uint32), rsmall.view(np.uint32))
        result["verified"] = bool(ok_big and all(shard_ok) and ok_small)
        result["verification"] = {"against": "oracle/o3dr_oracle.c (orc_run_frames) on the frames and poses of all ranks",
                                  "cloud_big_points_all_ranks": int(m1_total), "oracle_cloud_big_points": int(len(rbig)),
                                  "cloud_big_count_equal": bool(ok_big), "cloud_big_shard_sha1_equal": shard_ok,
                                  "merged_points": int(len(small)),
                                  "oracle_merged_points": int(len(rsmall)), "merged_bit_equal": bool(ok_small),
                                  "oracle_seconds": round(t_or, 2)}
        del rbig, all_disp, all_bgr

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.blur_kernel <= 1:
        nf = min(args.cpu_frames, F)
        v7, t7 = cpu_baseline(disp_h, bgr_h, poses_h, Q, args.voxel_size, args.jump_pixels, args.cpu_threads, nf, args.sor)
        n1 = min(100, F)
        v1, t1 = cpu_baseline(disp_h, bgr_h, poses_h, Q, args.voxel_size, args.jump_pixels, 1, n1, args.sor)
        result["cpu_baseline"] = {"value": round(v7, 3), "unit": "frames/s", "cores": args.cpu_threads, "kind": "port",
                                  "sample": f"first {nf} frames of the same workload through the C oracle (A6 per frame on "
                                            f"{args.cpu_threads} frame-parallel threads as pose.cpp:392-413, then the combined "
                                            f"merge), {t7:.1f} s; single thread: {v1:.3f} frames/s on {n1} frames",
                                  "single_thread_value": round(v1, 3), "host_cores": os.cpu_count()}
        if "sor_on" in result and "cpu_frames_per_sec" in result["sor_on"]:
            result["cpu_baseline"]["sor_on_value"] = result["sor_on"]["cpu_frames_per_sec"]
            result["cpu_baseline"]["sor_on_sample"] = (f"first {result['sor_on']['cpu_sample_frames']} frames with statistical outlier "
                                                       f"removal on, {args.cpu_threads} threads, {result['sor_on']['cpu_seconds']} s")
        # all cores of this host (frame-parallel part only scales; the merge is one thread, as in
        # the reference), plus what the host is
        try:
            n_all = max(1, min(len(os.sched_getaffinity(0)), nf))
            va, ta = cpu_baseline(disp_h, bgr_h, poses_h, Q, args.voxel_size, args.jump_pixels, n_all, nf, args.sor)
            model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "?")
            result["cpu_baseline"].update({"all_cores_value": round(va, 3), "all_cores_threads": n_all, "cpu_model": model})
        except Exception:
            pass
    if rank == 0:
        print(json.dumps(result), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if result.get("sor_on", {}).get("verified") is False:
        sys.exit("bench.py: with outlier removal on, the GPU clouds differ from the oracle's (see `sor_on` in the JSON line)")
    if result.get("verified") is False:
        sys.exit("bench.py: the GPU clouds differ from the oracle's (see `verification` in the JSON line)")


if __name__ == "__main__":
    main()
