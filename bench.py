#!/usr/bin/env python3
"""Throughput of the joint-bilateral hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process only spawns the N ranks, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one JointBilateralFilter::Process (K0 colour pre-smoothing + K1 joint bilateral filter)
over one batch of synthetic RGB-D frames that is already resident in HBM.  At N=1 the workload is
BASELINE config 2's filter on a batch of 640x480 frames (64 frames per GPU: at N=8 that is config 4,
512 frames sharded 8 ways; weak scaling, no data-path collective — frames are independent units;
the only exchange is one broadcast of the parameter block from rank 0).

One JSON line is printed by rank 0; besides the contract fields it carries
  roofline     — the dominant kernel (K1) against the HBM roofline: algorithmic 11 B/pixel
                 (4 B depth read + 3 B packed-BGR guide read + 4 B filtered write) x pixels per launch
                 / its average launch duration, measured with HIP events on the launch stream
                 inside the timed region.  K1 is instruction-issue bound, not HBM-bound, so the object also
                 carries `valu`: issue slots per second against the chip's peak (1024 SIMDs x clock / 4 cycles; a
                 transcendental takes two slots), with the per-wave instruction counts from the committed PMC
                 profile of exactly this code (profiles/pmc_bench.json, matched by a hash of the kernel sources)
                 and the launch time measured live; and `fhd_w19`: the same figures for the pass north_star's
                 roofline target names (32 x 1920x1080, window 19; BASELINE config 3);
  verified     — frame 0 of the TIMED output checked after the timed region, stage by stage (oracle.stage_check): the
                 stage build of the library (tools/hooks/libkde_hip_stage.so: same sources + dumps) must reproduce it to
                 the bit, its first-pass average is held to the float32 first-order bound of the binary64 average, and
                 the final value to 1e-4 against pass 2 evaluated in binary64 FROM that average; K0's u8 image exact;
  cpu_baseline — the CPU oracle (a scalar port of the CUDA kernels; the reference has no CPU path)
                 timed on this host on a bounded sample of the same workload (OpenMP over rows), and
  cpu_baseline_1t — the same on one thread;
  from_idle    — the same W + K steps measured first, from an idle GPU (no wake-up load): what the fixed W = 5 / K = 20
                 contract gives by itself; `value` is the steady-clock figure measured right after.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "Mpixels/sec joint-bilateral filtered (640x480 & 1080p); % HBM roofline"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
K1_BYTES_PER_PX = 11.0
CLK_GHZ = 2.4                  # max shader clock (same guide)
N_SIMD = 1024                  # 256 CUs x 4
SLOT_CYCLES = 4.0              # one wave64 VALU instruction per 4 cycles per SIMD (a transcendental: 8), the model
                               # DESIGN.md prices K1 with; profiles/r02_valu_microbench.txt has the measured costs


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-gpu", type=int, default=64)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--window", type=int, default=11, help="BASELINE 'radius=5' -> 2*5+1")
    ap.add_argument("--spatial-sigma", type=float, default=3.0)
    ap.add_argument("--color-sigma", type=float, default=7.65, help="sigma_r=0.03 of the 0..255 range")
    ap.add_argument("--depth-sigma", type=float, default=20.0)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--wakeup-ms", type=float, default=150.0,
                    help="untimed load before the W warm-up steps so that the GPU has left its idle clock level (0 = none)")
    ap.add_argument("--distinct-frames", type=int, default=8, help="distinct synthetic frames, tiled to the batch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the 1080p / reference-constant side measurements")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_bench.json"))
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed output")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the N > 1 path (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (rehearsal of the N > 1 path on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--first-frame", type=int, default=0, help="global index of the first frame (N = 1 runs of one shard of a larger batch)")
    ap.add_argument("--no-idle-leg", action="store_true", help="skip the from-idle measurement that precedes the headline (profiling runs)")
    ap.add_argument("--dump-frames", default="",
                    help="write this rank's input frames (all colour frames, then all depth frames, raw) for examples/shard_replay --frames-file")
    ap.add_argument("--dry-run", action="store_true",
                    help="everything but the GPU work: launch, process group, parameter broadcast, partition, reductions (CPU test of the N > 1 path)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: spawn the N ranks ourselves -- one process per GPU, started
    BEFORE anything touches a GPU (this parent never imports torch.cuda), with the environment torch.distributed.run
    would give them -- relay rank 0's single JSON line and return the worst exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # a rank that dies leaves the others waiting in a collective: give them a grace period, then end them (exact PIDs)
    failed_at = None
    while any(q.poll() is None for q in procs):
        if failed_at is None and any(q.poll() not in (None, 0) for q in procs):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30.0:
            for q in procs:
                if q.poll() is None:
                    q.kill()
        time.sleep(0.1)
    out0 = procs[0].stdout.read().decode(errors="replace")
    sys.stdout.write(out0)
    sys.stdout.flush()
    return max(abs(q.returncode) for q in procs)


def pmc_lookup(path, window, variant_name=None):
    """K1 entry of the committed PMC table (tools/pmc_report.py) for the kernel that actually ran: the template instance of
    `variant_name` ("w11-pk2-16x16-false-v4[-noelide]" / "w11-sc2-16x16-false"), the launch with the largest grid.
    Only used when the table was taken on exactly these kernel sources."""
    try:
        import importlib.util
        import re
        spec = importlib.util.spec_from_file_location("pmc_report", os.path.join(ROOT, "tools", "pmc_report.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        source_hash = mod.source_hash
        pj = json.load(open(path))
        if pj.get("kernel_source_sha16") != source_hash():
            return None, "profile is stale (kernel sources changed since tools/profile_round.sh ran)"
        pat = rf"jbf_(pk|fast)_kernel<{window},"
        m = re.match(r"w(\d+)-(pk|sc)(\d+)-(\d+)x(\d+)-(true|false)(-v[14])?(-noelide)?$", variant_name or "")
        if m:
            w, kind, npx, bx, by, cache, vl, noel = m.groups()
            if kind == "pk":     # jbf_pk_kernel<WIN, NP, BX, BY, CACHE, CSKIP, VL, ELIDE_ON>
                pat = (rf"jbf_pk_kernel<{w}, {npx}, {bx}, {by}, {cache}, (true|false), {'true' if vl == '-v4' else 'false'}, "
                       rf"{'false' if noel else 'true'}>")
            else:                # jbf_fast_kernel<WIN, PX, BX, BY, CACHE, CSKIP>
                pat = rf"jbf_fast_kernel<{w}, {npx}, {bx}, {by}, {cache}, (true|false)>"
        c = [k for k in pj["kernels"] if re.search(pat, k["kernel"])]
        if not c:
            return None, "no K1 entry for this kernel in the profile"
        return max(c, key=lambda k: k["grid"]), os.path.relpath(path, ROOT)
    except Exception as e:          # no profile: the live figures stand alone
        return None, f"no profile ({type(e).__name__})"


def k1_roofline(px_per_launch, k1_ms, entry, src):
    """HBM roofline (the contract's fields) + the VALU issue-slot roofline of one K1 launch"""
    achieved = K1_BYTES_PER_PX * px_per_launch / (k1_ms * 1e-3) / 1e9
    r = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "algorithmic_bytes_per_launch": K1_BYTES_PER_PX * px_per_launch, "avg_launch_ms": k1_ms}
    peak_slots = N_SIMD * CLK_GHZ * 1e9 / SLOT_CYCLES
    valu = {"peak_slots_per_s": peak_slots, "clk_ghz": CLK_GHZ, "achieved_slots_per_s": None, "frac": None, "profile": src,
            "definition": "slots per launch = (SQ_INSTS_VALU + SQ_INSTS_VALU_TRANS_F32) of the profiled launch of this grid "
                          "(wave-instructions; a transcendental takes two 4-cycle slots) / live launch time; "
                          "peak = 1024 SIMDs x clk / 4"}
    if entry:
        d, c = entry["derived"], entry["counters"]
        r["traffic"] = d.get("hbm_bytes")
        if "SQ_INSTS_VALU" in c:
            slots = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            valu.update(achieved_slots_per_s=slots / (k1_ms * 1e-3), frac=slots / (k1_ms * 1e-3) / peak_slots,
                        valu_insts_per_wave=d.get("valu_insts_per_wave"), trans_insts_per_wave=d.get("trans_insts_per_wave"),
                        waves=c.get("SQ_WAVES"), grid=entry["grid"], vgprs=entry.get("vgprs"),
                        pmc_busy_frac=d.get("valu_busy"), pmc_waves_per_simd=d.get("waves_per_simd"))
    r["valu"] = valu
    return r


def verify_frame0(args, p, synth, first_seed, out0, smooth0, variant):
    """frame 0 of the timed output, checked after the timed region (the checkers are never part of the timed path):
    K0 bytes against the oracle; K1 stage by stage -- tools/hooks/libkde_hip_stage.so (the product sources + dumps) must
    reproduce the timed output to the bit, its first-pass average is compared with binary64 within the float32 bound, and
    the final value with pass 2 evaluated in binary64 from that average (oracle.stage_check).  The float32 restatement
    with its envelope (round 2's bar) is reported next to it as a cross-check."""
    import ctypes
    from oracle import oracle as O
    from tools.hooks import stage
    O.build()
    O.set_threads(usable_cores())
    bgr, depth = synth.make_frame(first_seed, args.width, args.height)
    ref, sm, env = O.jbf_process(depth, bgr, p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, return_all=True)
    k0_exact = bool(np.array_equal(smooth0, sm))
    q = type(p)()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(q))
    q.presmooth = 0
    sout, savg, _ = stage.jbf_stage_run(q, depth[None], sm[None], variant)
    same = bool(stage.bits_equal(sout[0], out0))
    st = O.jbf_stage(depth, sm, p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, avg_in=savg[0])
    r = O.stage_check(out0, st, 1e-4)
    chk = O.parity_check(out0, ref, env, 1e-4)
    fl = env.flagged
    with np.errstate(invalid="ignore", divide="ignore"):
        width = np.where(fl & (env.hi > 0), (env.hi - env.lo) / np.maximum(np.abs(ref.astype(np.float64)), 1e-300), 0.0)
    return {"ok": bool(k0_exact and same and not r["bad"].any() and not chk["bad"].any()), "frame": 0, "pixels": r["n"],
            "k0_u8_exact": k0_exact, "stage_build_bit_identical": same,
            "stagewise": {"bad": int(r["bad"].sum()), "band_pixels": r["band"], "band_frac": r["band_frac"],
                          "max_rel_err_strict": r["max_rel_strict"], "avg_checked": r["avg_checked"],
                          "avg_err_over_bound_p50_p99_max": [r["avg_bound_frac_p50"], r["avg_bound_frac_p99"], r["avg_bound_frac_max"]],
                          "avg_bound_rel_p50_max": [r["avg_tol_p50"], r["avg_tol_max"]],
                          "band_width_p50_p99_max": [r["band_width_p50"], r["band_width_p99"], r["band_width_max"]]},
            "float32_restatement_crosscheck": {"bad": int(chk["bad"].sum()), "flagged": chk["flagged"], "flagged_band": chk["band"],
                                               "flagged_cond": chk["cond"], "max_rel_err_unflagged": chk["max_rel_unflagged"],
                                               "max_rel_err_flagged_vs_f32": chk["max_rel_flagged"],
                                               "envelope_width_p50": float(np.percentile(width[fl], 50)) if fl.any() else 0.0,
                                               "envelope_width_p99": float(np.percentile(width[fl], 99)) if fl.any() else 0.0,
                                               "envelope_width_max": float(width.max()) if fl.any() else 0.0},
            "bar": "K0 bytes exact; stage build == timed output to the bit; first-pass average within its float32 first-order bound "
                   "of the binary64 average; every pixel with no tap on a Q1 decision at that average: identical zero mask and <= 1e-4 "
                   "against pass 2 evaluated in binary64 from it; BAND pixels inside the interval of both outcomes"}


def make_inputs(synth, torch, first_seed, n, w, h, distinct):
    distinct = max(1, min(distinct, n))
    bgr, depth = synth.make_batch(first_seed, distinct, w, h)
    reps = -(-n // distinct)
    bgr = np.tile(bgr, (reps, 1, 1, 1))[:n]
    depth = np.tile(depth, (reps, 1, 1))[:n]
    return torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()


def time_steps(torch, jbf, depth, color, smooth, out, steps, warmup, barrier, wakeup_ms=0.0):
    """device wake-up, W untimed + exactly K timed steps; returns (wall seconds, K0 ms list, K1 ms list, wake-up steps)."""
    def step(evs=None):
        if evs:
            evs[0].record()
        jbf.presmooth_batch(color, smooth)          # K0  (== what kde_jbf_process_batch launches)
        if evs:
            evs[1].record()
        jbf.filter_batch(depth, smooth, out)        # K1
        if evs:
            evs[2].record()

    barrier()                                       # also brings the RCCL communicator up before anything is timed
    # Device wake-up, before the W warm-up steps and outside every timed region: an idle MI355X sits at its lowest
    # clock level and needs ~100 ms of load to reach the clock it then holds; with W = 5 (6.5 ms) the K timed steps
    # would otherwise measure that ramp (first step 1.27 ms, last 1.11 ms) instead of the kernel.
    woke = 0
    t_w = time.perf_counter()
    while wakeup_ms > 0 and (time.perf_counter() - t_w) * 1e3 < wakeup_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        woke += 10
    for _ in range(warmup):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0                   # this rank's K steps, from the common start; MAX over ranks is taken
    barrier()                                       # by the caller (the collective's own latency is not part of a step)
    k0 = [e[0].elapsed_time(e[1]) for e in evs]
    k1 = [e[1].elapsed_time(e[2]) for e in evs]
    return dt, k0, k1, woke


def usable_cores():
    """threads the CPU leg may really use: the cgroup CPU quota if one is set, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    # a one-GPU box of the pool is granted a 16-CPU share of its 256-CPU host (the affinity mask still shows
    # all of them; 256 OpenMP threads on that share ran 10x slower than 16)
    return min(n, int(os.environ.get("KDE_CPU_THREADS", "16")))


def cpu_baseline(args, synth, seconds, threads=None):
    from oracle import oracle as O
    O.build()
    cores = usable_cores() if threads is None else threads
    O.set_threads(cores)
    frames = [synth.make_frame(s, args.width, args.height) for s in range(4)]       # distinct frames, cycled
    O.jbf_process(frames[0][1], frames[0][0], args.window, args.spatial_sigma, args.color_sigma, args.depth_sigma)   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        bgr, depth = frames[n % len(frames)]
        O.jbf_process(depth, bgr, args.window, args.spatial_sigma, args.color_sigma, args.depth_sigma)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds and n >= 2:
            break
    return {"value": n * args.width * args.height / el / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{n} x ({args.width}x{args.height}) frames (4 distinct, cycled) of the same Process (K0+K1, window {args.window}) "
                      f"through oracle/kde_oracle.c, " + (f"OpenMP over rows on {cores} threads" if cores > 1 else "one thread") + f", {el:.1f} s"}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))            # the parent never touches a GPU
    # RCCL prints its version banner / warnings on the C-level stdout: keep the real stdout for the ONE JSON
    # line of the contract and send everything else that writes to fd 1 to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from kinectdepthmapenhancement_amd import sharding, synth
    from kinectdepthmapenhancement_amd._native import JbfParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} != --gpus {args.gpus}")
    if args.share_device and args.backend == "nccl" and world > 1:
        raise SystemExit("--share-device needs --backend gloo (RCCL wants one device per rank)")
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run or by launch_ranks (also for N = 1)
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl" and not args.dry_run:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    barrier = dist.barrier if use_dist else (lambda: None)

    # ---- parameter block: rank 0 decides, everyone receives the same bytes --------------------------
    if args.dry_run:
        p = JbfParams(args.window, args.spatial_sigma, args.color_sigma, args.depth_sigma, 1, 5, 30.0, 30.0)
        r = args.window // 2
        yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
        host_table = lambda q: np.exp(-(xx * xx + yy * yy) / (2.0 * q.spatial_sigma ** 2)).astype(np.float32)
    else:
        from kinectdepthmapenhancement_amd import filters
        p = filters.JointBilateralFilter.default_params()
        p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = args.window, args.spatial_sigma, args.color_sigma, args.depth_sigma
        host_table = lambda q: filters.JointBilateralFilter(8, 8, q).spatial_table()
    blk = sharding.pack_params(p, table=host_table(p)) if rank == 0 else np.zeros(sharding.BLOCK_LEN)
    blk = sharding.broadcast_params(blk)
    p, _, _, _, table0 = sharding.unpack_params(blk)
    replicas_only = False
    if not np.array_equal(host_table(p), table0):
        replicas_only = True        # would mean ranks disagree on the host-computed table: flag the run

    # ---- this rank's shard of the global batch ------------------------------------------------------
    total_frames = args.frames_per_gpu * world
    first, count = sharding.partition(total_frames, world)[rank]
    first += args.first_frame
    W, H = args.width, args.height
    if args.dry_run:
        barrier()
        dt = sharding.allreduce_max(1e-3 * (rank + 1))
        checksum = sharding.allreduce_sum([float(first), float(count)])
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": True, "max_dt_over_ranks": dt, "replicas_only": replicas_only,
                              "checksum": {"sum_first_frames": checksum[0], "frames": int(checksum[1])},
                              "config": {"window": p.window_size, "frames_per_gpu": count}}), file=real_stdout, flush=True)
        if use_dist:
            dist.destroy_process_group()
        return
    color, depth = make_inputs(synth, torch, first, count, W, H, args.distinct_frames)
    if args.dump_frames and rank == 0:
        with open(args.dump_frames, "wb") as f:
            f.write(color.cpu().numpy().tobytes())
            f.write(depth.cpu().numpy().tobytes())
    smooth = torch.empty_like(color)
    out = torch.empty_like(depth)
    jbf = filters.JointBilateralFilter(W, H, p, max_batch=count)
    if args.variant >= 0:
        jbf.set_variant(args.variant)

    # from an idle GPU first (the W + K contract by itself), then at the clock the GPU holds under load (the headline)
    idle = None
    if args.wakeup_ms > 0 and not args.no_idle_leg:
        dt_i, _, k1_i, _ = time_steps(torch, jbf, depth, color, smooth, out, args.steps, args.warmup, barrier, 0.0)
        dt_i = sharding.allreduce_max(dt_i)
        idle = {"value": total_frames * W * H * args.steps / dt_i / 1e6, "ms_per_step": dt_i / args.steps * 1e3,
                "k1_launch_ms_first_median_last": [float(k1_i[0]), float(np.median(k1_i)), float(k1_i[-1])],
                "note": "same W warm-up + K timed steps started from an idle GPU (lowest clock level), measured before the headline"}
    dt, k0_ms, k1_ms, woke = time_steps(torch, jbf, depth, color, smooth, out, args.steps, args.warmup, barrier, args.wakeup_ms)
    dt = sharding.allreduce_max(dt)
    checksum = sharding.allreduce_sum([float(out.double().sum().item()), float(count)])

    if rank == 0:
        px_per_launch = count * W * H
        k1_avg_ms = float(np.mean(k1_ms))
        names = filters.JointBilateralFilter.variants()
        vname = names[args.variant] if args.variant >= 0 else next((nm for nm in names[1:] if nm.startswith(f"w{p.window_size}-")), None)
        entry, src = pmc_lookup(args.pmc_json, p.window_size, vname)
        roof = {"bound": "hbm", "bound_measured": "valu-issue",
                "limiter": "valu-issue (K1 does 2 exp + ~30 flops per tap against 11 B/pixel; see roofline.valu)",
                "kernel": "K1 joint_bilateral_filtering"}
        roof.update(k1_roofline(px_per_launch, k1_avg_ms, entry, src))
        roof["k0_avg_launch_ms"] = float(np.mean(k0_ms))
        roof["launch_ms_first_min_max"] = [float(k1_ms[0]), float(np.min(k1_ms)), float(np.max(k1_ms))]   # clock ramp shows here
        roof["launch_ms"] = {"mean": k1_avg_ms, "median": float(np.median(k1_ms)), "min": float(np.min(k1_ms)),
                             "max": float(np.max(k1_ms)), "first": float(k1_ms[0])}
        res = {
            "metric": METRIC,
            "value": total_frames * W * H * args.steps / dt / 1e6,
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "wakeup_steps_before_warmup": woke,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"JointBilateralFilter::Process (K0 pre-smooth 5/30/30 + K1) on {total_frames} x {W}x{H} "
                            f"synthetic RGB-D frames ({count} per GPU), window {p.window_size} (radius {p.window_size // 2}), "
                            f"sigma_s {p.spatial_sigma:g} px, sigma_r {p.color_sigma:g}/255, sigma_d {p.depth_sigma:g} mm",
                "frames_per_gpu": count, "width": W, "height": H, "window": p.window_size,
                "sharding": f"contiguous frame blocks x{world}, params broadcast from rank 0 ({args.backend if use_dist else 'single process'}"
                            + (", all ranks on cuda:0" if args.share_device else "") + ")" + (" [REPLICAS ONLY]" if replicas_only else ""),
                "kernel_variant": (names[args.variant] if args.variant >= 0 else f"auto ({vname})"),
            },
            "roofline": roof,
            "checksum": {"sum_filtered_mm": checksum[0], "frames": int(checksum[1])},
        }
        if idle:
            res["from_idle"] = idle
        if not args.no_verify:
            res["verified"] = verify_frame0(args, p, synth, first, out[0].cpu().numpy(), smooth[0].cpu().numpy(), args.variant)
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(args, synth, args.cpu_seconds)
            res["cpu_baseline_1t"] = cpu_baseline(args, synth, max(3.0, args.cpu_seconds / 2), threads=1)
        if world == 1 and not args.no_extra:
            res["also"] = side_measurements(torch, filters, synth, args)
            res["roofline"]["fhd_w19"] = res["also"].pop("fhd_w19_config3")
        print(json.dumps(res), file=real_stdout, flush=True)
    if use_dist:
        dist.destroy_process_group()


def side_measurements(torch, filters, synth, args):
    """not the headline: K1 alone at 1080p / window 19 (BASELINE config 3) and with the reference's
    compile-time constants (window 5, sigma 70/50/20) on the VGA batch; same event timing."""
    out = {}

    def run(name, w, h, n, distinct, window, ss, cs, ds):
        p = filters.JointBilateralFilter.default_params()
        p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = window, ss, cs, ds
        color, depth = make_inputs(synth, torch, 1000, n, w, h, distinct)
        smooth, res = torch.empty_like(color), torch.empty_like(depth)
        jbf = filters.JointBilateralFilter(w, h, p, max_batch=n)
        if args.variant >= 0:
            jbf.set_variant(args.variant)
        steps = max(3, min(args.steps, 10))
        dt, k0, k1, _ = time_steps(torch, jbf, depth, color, smooth, res, steps, 2, lambda: None)
        px = n * w * h
        k1m = float(np.mean(k1))
        names = filters.JointBilateralFilter.variants()
        vname = names[args.variant] if args.variant >= 0 else next((nm for nm in names[1:] if nm.startswith(f"w{window}-")), None)
        entry, src = pmc_lookup(args.pmc_json, window, vname)
        out[name] = {"workload": f"K0 + K1 on {n} x {w}x{h}, window {window}, sigma {ss:g}/{cs:g}/{ds:g}",
                     "frames": n, "width": w, "height": h, "window": window,
                     "process_mpix_s": px * steps / dt / 1e6, "k1_mpix_s": px / (k1m * 1e-3) / 1e6,
                     "k0_avg_launch_ms": float(np.mean(k0)), "bound": "hbm", "limiter": "valu-issue"}
        out[name].update(k1_roofline(px, k1m, entry, src))

    # BASELINE config 3 as SURVEY 8(d) sizes it: 32 frames = 730 MB of algorithmic traffic, beyond the 256 MB Infinity Cache
    run("fhd_w19_config3", 1920, 1080, 32, 2, 19, 3.0, 7.65, 20.0)
    run("vga_reference_constants_w5", 640, 480, 64, 8, 5, 70.0, 50.0, 20.0)
    # ---- the headline's dependence on content (tile-level rule elision fires on smooth tiles only): K1 alone on the
    # synthetic frames and on the reference's own colour frame (input/color.jpg decode, textured) tiled to the batch,
    # each with the default kernel and with its "-noelide" twin (every tile runs the full-rule body: the floor)
    try:
        from PIL import Image
        fix = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "color_640x480.png")).convert("RGB"))[..., ::-1])
    except Exception:
        fix = None
    names = filters.JointBilateralFilter.variants()
    p11 = filters.JointBilateralFilter.default_params()
    p11.window_size, p11.spatial_sigma, p11.color_sigma, p11.depth_sigma, p11.presmooth = 11, 3.0, 7.65, 20.0, 0
    n64 = 64
    syn_c, syn_d = make_inputs(synth, torch, 0, n64, 640, 480, 8)
    content = {"synthetic": syn_c}
    if fix is not None:
        content["reference_colour_frame_x64"] = torch.from_numpy(fix).cuda()[None].repeat(n64, 1, 1, 1).contiguous()
    res11 = torch.empty_like(syn_d)
    legs = {}
    pre = filters.JointBilateralFilter(640, 480, max_batch=n64)            # K0 with the reference's 5/30/30, as in the headline step
    for cname, col in content.items():
        guide = torch.empty_like(col)
        pre.presmooth_batch(col, guide)
        js = {}
        for vname in ("auto", "w11-pk2-16x16-false-v4-noelide"):
            if vname != "auto" and vname not in names:
                continue
            j = filters.JointBilateralFilter(640, 480, p11, max_batch=n64)
            if vname != "auto":
                j.set_variant(names.index(vname))
            js["default" if vname == "auto" else "noelide"] = j
        times = {k: [] for k in js}
        for rnd in range(12):                                              # interleaved rounds: clock / thermal drift cancels
            for k, j in js.items():
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                j.filter_batch(syn_d, guide, res11)
                b_.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[k].append(a_.elapsed_time(b_))
        for k, tl in times.items():
            ms = float(np.median(tl))
            legs[f"{cname}/{k}"] = {"k1_ms": ms, "k1_mpix_s": n64 * 640 * 480 / ms / 1e3}
        for j in js.values():
            j.close()
    pre.close()
    out["k1_w11_content_dependence"] = {"workload": "K1 alone on the K0-smoothed guide (as in the headline step), 64 x 640x480, window 11, sigma "
                                                    "3/7.65/20; default kernel (tile-level rule elision) and its -noelide twin (every tile runs the "
                                                    "full-rule body) in interleaved rounds, median of 10", **legs}
    del content, syn_c, syn_d, res11

    # empirical HBM ceiling: float4 copy of 1 GiB (read + write)
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    import importlib.util
    spec = importlib.util.spec_from_file_location("kde_hooks", os.path.join(ROOT, "tools", "hooks", "hooks.py"))
    hooks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hooks)          # tools/hooks/libkde_hooks.so: measurement helper, not in the product library
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        hooks.hbm_copy(a, b, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        hooks.hbm_copy(a, b, st)
    e1.record()
    torch.cuda.synchronize()
    out["float4_copy_GBs"] = 2 * 4 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b

    # BASELINE config 5: full chain on one 1080p frame (projectiveToReal -> JBF.Process -> projectiveToReal ->
    # RGBF.Process fed with the JBF output), and the HBM-bound feeder projectiveToReal on a 32-frame batch
    W, H = 1920, 1080
    bgr, depth = synth.make_frame(2000, W, H)
    K = synth.intrinsics(W, H)
    color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
    conv = filters.DimensionConvertor()
    conv.setCameraParameters(K, W, H)
    jbf = filters.JointBilateralFilter(W, H)
    rg = filters.RegionGrowingBilateralFilter(W, H)
    rg.SetParametor(15, 20, K)
    pts = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    filt = jbf.getFiltered_Device()

    def chain():
        conv.projectiveToReal(d, pts)
        jbf.Process(d, color)
        conv.projectiveToReal(filt, pts)
        rg.Process(filt, pts, color)

    def timed(fn, iters=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms = timed(chain)
    out["fhd_full_chain_config5"] = {"ms_per_frame": ms, "mpix_s": W * H / ms / 1e3, "rows": 15, "cols": 20}
    # the same chain on a BATCH of frames (north_star's unit): every stage takes the whole batch per launch
    # (kde_jbf_process_batch, kde_rgbf_process_batch), next to the same frames pushed through the single-frame calls
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_chain_batch", os.path.join(ROOT, "tools", "bench_chain_batch.py"))
    bcb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bcb)
    out["vga_chain_batch64"] = bcb.run(torch, filters, synth, 640, 480, 64, 6)
    out["fhd_chain_batch8"] = bcb.run(torch, filters, synth, 1920, 1080, 8, 6)
    db = d[None].repeat(32, 1, 1).contiguous()
    pb = torch.empty((32, H, W, 3), dtype=torch.float32, device="cuda")
    ms = timed(lambda: conv.projectiveToReal(db, pb))
    out["projectiveToReal_32xfhd"] = {"ms": ms, "GBs": 16.0 * 32 * W * H / ms / 1e6, "hbm_frac": 16.0 * 32 * W * H / ms / 1e6 / HBM_PEAK_GBS}
    return out


if __name__ == "__main__":
    main()
