#!/usr/bin/env python3
"""Throughput of the joint-bilateral hot path on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: this process only spawns the N ranks, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one JointBilateralFilter::Process (K0 colour pre-smoothing + K1 joint bilateral filter)
over one batch of synthetic RGB-D frames that is already resident in HBM, made through the boundary entry
point a caller uses: ONE kde_jbf_process_batch call per step (the drop-in of JointBilateralFilter::Process,
JointBilateralFilter.cu:283-290), two HIP events around each step.  The K0 / K1 split that `roofline` reads
is measured in a second, equally long leg right after (kde_jbf_presmooth_batch + kde_jbf_filter_batch, the
two launches the boundary call makes, three events per step).  At N=1 the workload is
BASELINE config 2's filter on a batch of 640x480 frames (64 frames per GPU: at N=8 that is config 4,
512 frames sharded 8 ways; weak scaling, no data-path collective — frames are independent units;
the only exchange is one broadcast of the parameter block from rank 0).

One JSON line is printed by rank 0; besides the contract fields it carries
  roofline     — the dominant kernel (K1) against the HBM roofline: algorithmic 11 B/pixel
                 (4 B depth read + 3 B packed-BGR guide read + 4 B filtered write) x pixels per launch
                 / its average launch duration, measured with HIP events on the launch stream
                 (the split leg).  K1 is instruction-issue bound, not HBM-bound, so the object also carries
                 `valu.mix_ceiling`: the shader cycles the kernel's VALU instructions alone need -- its own mix, each opcode at
                 the issue rate of its MEASURED class (2 / 4 / 8 cycles: tools/valu_microbench -> profiles/valu_costs.json;
                 tools/valu_mix.py on the committed PMC counts of exactly this code, profiles/pmc_bench.json, matched by a
                 hash of the kernel sources) -- over the cycles of the same profiled launch; and `fhd_w19`: the same figures for the pass north_star's
                 roofline target names (32 x 1920x1080, window 19; BASELINE config 3), run by every rank, with
                 `k1_mpix_s_noelide` = its data-independent floor;
  verified     — frame 0 of the TIMED output checked after the timed region, stage by stage (oracle.stage_check): the
                 stage build of the library (tools/hooks/libkde_hip_stage.so: same sources + dumps) must reproduce it to
                 the bit, its first-pass average is held to the float32 first-order bound of the binary64 average, and
                 the final value to 1e-4 against pass 2 evaluated in binary64 FROM that average; K0's u8 image exact;
  cpu_baseline — the CPU oracle (a scalar port of the CUDA kernels; the reference has no CPU path)
                 timed on this host on a bounded sample of the same workload (OpenMP over rows), and
  cpu_baseline_1t — the same on one thread;
  from_idle    — the same W + K steps measured first, from an idle GPU (no wake-up load): what the fixed W = 5 / K = 20
                 contract gives by itself; `value` is the steady-clock figure measured right after;
  ranks_seen / devices / rccl / replicas_only — who took part (an all-reduce of 1; rank, pid, arch and PCI address of every
                 rank's GPU) and over what: RCCL when it came up on every rank, else the in-process gloo fallback, flagged;
  also         — side legs: at every N the config-5 chain on 64 x 640x480 per GPU through the batched entry points (run by
                 every rank, reduced like the headline); at N = 1 the reference constants, K1's dependence on content at
                 windows 11 and 19, this chip's copy ceiling, the single-frame chain, the feeder.
The barriers around timed regions are host-side (gloo) and the GPU is kept busy through the start barrier (time_steps):
an idle gap of a few milliseconds in front of the K timed steps makes an MI355X drop its clock.
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
SLOT_CYCLES = 4.0              # r01-r03's nominal slot model, kept as a comparison figure only: gfx950 issues v_mul / v_add / v_fma_f32 in
                               # 2 cycles, so "4 cycles per instruction" is not a ceiling; roofline.valu.mix_ceiling prices the real mix


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-gpu", type=int, default=64)
    ap.add_argument("--total-frames", type=int, default=0,
                    help="strong scaling: a FIXED batch (BASELINE config 4 as written: 512) cut into ceil(T/N) frames per GPU; "
                         "0 = weak scaling with --frames-per-gpu frames on every GPU (the default)")
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
    ap.add_argument("--no-extra", action="store_true", help="skip the 1080p / chain / reference-constant side measurements")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_bench.json"))
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed output")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the N > 1 path (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (rehearsal of the N > 1 path on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--force-rccl-failure", action="store_true",
                    help="make the RCCL bring-up raise on every rank: exercises SURVEY 8(e)'s in-process fallback (gloo, [REPLICAS ONLY])")
    ap.add_argument("--rccl-timeout", type=float, default=60.0,
                    help="deadline in seconds of each RCCL bring-up step (the disposable probe processes, then new_group + the first "
                         "all-reduce in the rank itself); past it the run goes on over gloo, flagged [REPLICAS ONLY]")
    ap.add_argument("--no-rccl-probe", action="store_true", help="skip the disposable probe processes (the in-process deadline still holds)")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("KDE_BENCH_LAUNCH_TIMEOUT", "1500")),
                    help="N > 1 without a launcher: seconds after which the parent ends every rank it started (exact PIDs)")
    ap.add_argument("--first-frame", type=int, default=0, help="global index of the first frame (N = 1 runs of one shard of a larger batch)")
    ap.add_argument("--no-idle-leg", action="store_true", help="skip the from-idle measurement that precedes the headline (profiling runs)")
    ap.add_argument("--dump-frames", default="",
                    help="write this rank's input frames (all colour frames, then all depth frames, raw) for examples/shard_replay --frames-file")
    ap.add_argument("--dry-run", action="store_true",
                    help="everything but the GPU work: launch, process groups, fallback, parameter broadcast, partition, the reductions of "
                         "every leg with stand-in timings (CPU test of the N > 1 path)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: spawn the N ranks ourselves -- one process per GPU, started
    BEFORE anything touches a GPU (this parent never imports torch.cuda), with the environment torch.distributed.run
    would give them -- relay rank 0's single JSON line and return the worst exit code.

    HSA_ENABLE_IPC_MODE_LEGACY=0 (kept if the caller set it, as this image and the GPU pool do): the pool's host driver only
    supports dmabuf IPC; with the legacy mode RCCL's intra-node P2P set-up -- which exports its buffers to the other ranks
    through HIP IPC handles -- fails in hipIpcGetMemHandle with "invalid argument".  The data path never uses IPC (frames
    are not exchanged); only RCCL's own bring-up for the parameter broadcast does."""
    import threading
    # RCCL is tried first in N disposable probe processes (gloo rendezvous -> new_group("nccl") -> one all-reduce -> exit)
    # under a deadline: a bring-up that hangs on one rank costs --rccl-timeout seconds and a flagged line, never the record.
    # The verdict travels to the ranks in the environment; with a bad verdict they never touch RCCL.
    verdict = None
    if (args.backend == "nccl" and not args.no_rccl_probe and not args.force_rccl_failure
            and (not args.dry_run or os.environ.get("KDE_RCCL_PROBE_TEST"))):
        import importlib.util       # the module file itself: the package import would pull torch into this parent
        spec = importlib.util.spec_from_file_location("kde_rccl_probe", os.path.join(ROOT, "kinectdepthmapenhancement_amd", "rccl_probe.py"))
        rccl_probe = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rccl_probe)
        verdict = rccl_probe.run_probes(args.gpus, args.rccl_timeout, args.share_device)
        print(f"bench.py: RCCL probe x{args.gpus}: {'ok' if verdict['ok'] else 'FAILED -- ' + verdict['reason']} ({verdict['seconds']} s)",
              file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]      # released here and re-bound by rank 0's store: a stolen port fails the launch loudly (rc != 0)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if verdict is not None:
            env["KDE_RCCL_PROBE_VERDICT"] = json.dumps(verdict)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)    # drained while the ranks run:
    reader.start()                                                                                    # a full pipe never blocks rank 0
    # a rank that dies leaves the others waiting in a collective: give them a grace period, then end them (exact PIDs);
    # ranks that all hang are ended at the overall deadline
    deadline, failed_at, timed_out = time.time() + args.launch_timeout, None, False
    while any(q.poll() is None for q in procs):
        now = time.time()
        if failed_at is None and any(q.poll() not in (None, 0) for q in procs):
            failed_at = now
        if (failed_at is not None and now - failed_at > 30.0) or now > deadline:
            timed_out = timed_out or now > deadline
            for q in procs:
                if q.poll() is None:
                    q.kill()
        time.sleep(0.1)
    reader.join(10.0)
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    rc = max(abs(q.returncode) for q in procs)
    if timed_out:
        print(f"bench.py: the ranks did not finish within --launch-timeout {args.launch_timeout:g} s and were ended", file=sys.stderr)
        rc = rc or 124
    return rc


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
            if kind == "pk":     # jbf_pk_kernel<WIN, NP, BX, BY, CACHE, CSKIP, VL, ELIDE_ON, CR> (CR = 0 or WIN in the product build)
                pat = (rf"jbf_pk_kernel<{w}, {npx}, {bx}, {by}, {cache}, (true|false), {'true' if vl == '-v4' else 'false'}, "
                       rf"{'false' if noel else 'true'}(, \d+)?>")
            else:                # jbf_fast_kernel<WIN, PX, BX, BY, CACHE, CSKIP>
                pat = rf"jbf_fast_kernel<{w}, {npx}, {bx}, {by}, {cache}, (true|false)>"
        c = [k for k in pj["kernels"] if re.search(pat, k["kernel"])]
        if not c:
            return None, "no K1 entry for this kernel in the profile"
        return max(c, key=lambda k: k["grid"]), os.path.relpath(path, ROOT)
    except Exception as e:          # no profile: the live figures stand alone
        return None, f"no profile ({type(e).__name__})"


def k1_roofline(px_per_launch, k1_ms, entry, src):
    """HBM roofline (the contract's fields) + the VALU-issue ceiling of one K1 launch: the kernel's own instruction mix priced
    with the measured issue cost of every opcode (tools/valu_mix.py on the committed PMC counts of exactly this code; costs from
    tools/valu_microbench, profiles/valu_costs.json) against the launch time measured live."""
    achieved = K1_BYTES_PER_PX * px_per_launch / (k1_ms * 1e-3) / 1e9
    r = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "algorithmic_bytes_per_launch": K1_BYTES_PER_PX * px_per_launch, "avg_launch_ms": k1_ms}
    valu = {"mix_ceiling": None, "profile": src,
            "definition": "mix_ceiling = bare_stream_cycles / launch cycles (GRBM_GUI_ACTIVE / 8), both of the committed PMC run of exactly this "
                          "code: bare_stream_cycles = sum over VALU opcodes of (instructions per wave x measured issue cost in shader cycles, "
                          "rounded to its class: 2, 4 or 8 cycles) x waves / 1024 SIMDs -- an ESTIMATE of what the kernel's VALU instructions "
                          "alone take with every SIMD issuing back to back (the class costs are measured costs of pure streams, not lower "
                          "bounds); both terms are cycle counts of one run, so the fraction does not depend on the clock"}
    if entry:
        d, c, mix = entry["derived"], entry["counters"], entry.get("mix")
        r["traffic"] = d.get("hbm_bytes")
        if mix:
            bare_ms = mix["bare_stream_ns"] * 1e-6
            valu.update(mix_ceiling=d.get("valu_cycles_frac"), bare_stream_cycles=mix.get("bare_stream_cycles"), launch_cycles_profiled=d.get("cycles"),
                        mean_cycles_per_instruction=mix.get("mean_cycles_per_instruction"),
                        bare_stream_ms_at_microbench_clock=bare_ms,
                        class_fractions=mix["class_fractions"], not_in_cost_table_frac=mix["not_in_cost_table_frac"])
        if "SQ_INSTS_VALU" in c:
            slots = c["SQ_INSTS_VALU"] + c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            valu.update(valu_insts_per_wave=d.get("valu_insts_per_wave"), trans_insts_per_wave=d.get("trans_insts_per_wave"),
                        waves=c.get("SQ_WAVES"), grid=entry["grid"], vgprs=entry.get("vgprs"), pmc_waves_per_simd=d.get("waves_per_simd"),
                        nominal_4cycle_slot_frac=slots / (k1_ms * 1e-3) / (N_SIMD * CLK_GHZ * 1e9 / SLOT_CYCLES),
                        nominal_4cycle_slot_note="r01-r03's model (every VALU instruction one 4-cycle slot, a transcendental two, at 2.4 GHz): "
                                                 "not a ceiling for kernels with 2-cycle opcodes; kept for comparison")
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
    chk = O.parity_check(out0, ref, env, 1e-4, grid=r["grid_map"])
    cen = chk["census"]
    fl = env.flagged
    with np.errstate(invalid="ignore", divide="ignore"):
        width = np.where(fl & (env.hi > 0), (env.hi - env.lo) / np.maximum(np.abs(ref.astype(np.float64)), 1e-300), 0.0)
    return {"ok": bool(k0_exact and same and not r["bad"].any() and not chk["bad"].any()), "frame": 0, "pixels": r["n"],
            "k0_u8_exact": k0_exact, "stage_build_bit_identical": same,
            "stagewise": {"bad": int(r["bad"].sum()), "band_pixels": r["band"], "band_frac": r["band_frac"],
                          "band_decision_pixels": r["band_decision"], "grid_pixels": r["grid"], "avg_checked_of_strict": r["avg_checked_of_strict"],
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
            "census_vs_float32": {**cen, "what": "END TO END against the float32 restatement of the reference kernels, every pixel of frame 0 counted "
                                                  "whatever its class: pixels more than 1e-4 (relative) from the float32 value, pixels whose zero mask "
                                                  "differs, and the same over the denormal-grid class (GRID).  Ceilings: tests/test_gpu_fullsize.py"},
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


def time_steps(torch, jbf, depth, color, smooth, out, steps, warmup, barrier, wakeup_ms=0.0, split=False, keep_busy_ms=0.0):
    """device wake-up, W untimed + exactly K timed steps.
    split=False: a step is ONE boundary call, kde_jbf_process_batch (JointBilateralFilter::Process on the batch), with an event
                 before and after it;
    split=True : the two launches that call makes, issued separately (kde_jbf_presmooth_batch = K0, kde_jbf_filter_batch = K1)
                 with an event between them: the per-kernel times `roofline` reads.
    keep_busy_ms: untimed steps worth that much GPU time are enqueued right before the barrier that precedes the timed region,
                 so that the GPU is still working while the host waits for the other ranks: the synchronize() that follows the
                 barrier then returns microseconds before the first timed launch.  Without it the GPU idles for the length of
                 the barrier (milliseconds over TCP or RCCL), drops its clock, and the first timed steps measure the ramp
                 (1.45 ms instead of 1.19: tools/exp_dist_overhead.sh).  0 = none (the from-idle leg).
    returns (wall seconds of the K steps, per-step ms list [K0 + K1 when split], K0 ms list | None, K1 ms list | None,
             wake-up steps, untimed steps enqueued before the barrier)."""
    def step(evs=None):
        if evs:
            evs[0].record()
        if split:
            jbf.presmooth_batch(color, smooth)          # K0
            if evs:
                evs[1].record()
            jbf.filter_batch(depth, smooth, out)        # K1
        else:
            jbf.process_batch(depth, color, out)        # the drop-in of JointBilateralFilter::Process (.cu:283-290)
        if evs:
            evs[-1].record()

    barrier()
    # Device wake-up, before the W warm-up steps and outside every timed region: an idle MI355X sits at its lowest
    # clock level and needs ~100 ms of load to reach the clock it then holds; with W = 5 (6.5 ms) the K timed steps
    # would otherwise measure that ramp (first step 1.27 ms, last 1.11 ms) instead of the kernel.
    woke, est_ms = 0, None
    t_w = time.perf_counter()
    while wakeup_ms > 0 and (time.perf_counter() - t_w) * 1e3 < wakeup_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        woke += 10
    if woke:
        est_ms = (time.perf_counter() - t_w) * 1e3 / woke
    elif keep_busy_ms > 0:                          # no wake-up loop to take the step time from: one probe step
        t_p = time.perf_counter()
        step()
        torch.cuda.synchronize()
        est_ms = (time.perf_counter() - t_p) * 1e3
    for _ in range(warmup):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3 if split else 2)] for _ in range(steps)]
    filler = 0
    if keep_busy_ms > 0:
        filler = max(2, min(400, int(keep_busy_ms / max(est_ms, 0.02)) + 1))
        for _ in range(filler):
            step()                                  # untimed: keeps the GPU busy while the host is in the barrier
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0                   # this rank's K steps, from the common start; MAX over ranks is taken
    barrier()                                       # by the caller (the collective's own latency is not part of a step)
    step_ms = [e[0].elapsed_time(e[-1]) for e in evs]
    k0 = [e[0].elapsed_time(e[1]) for e in evs] if split else None
    k1 = [e[1].elapsed_time(e[2]) for e in evs] if split else None
    return dt, step_ms, k0, k1, woke, filler


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


def device_identity(rank, local_rank, dry_run):
    """what tells this rank's GPU apart from the others of the node (the line shows N distinct PCI addresses for N GPUs)"""
    ident = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid()}
    if not dry_run:
        import ctypes
        from kinectdepthmapenhancement_amd._native import check, lib
        arch, bus, cus = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64), ctypes.c_int(0)
        check(lib().kde_device_info(arch, 64, ctypes.byref(cus)))
        check(lib().kde_device_pci_bus_id(bus, 64))
        ident.update(arch=arch.value.decode(), pci_bus_id=bus.value.decode(), cu_count=cus.value)
    return ident


def reduce_shard_legs(per_rank, world):
    """the side legs every rank ran on its own GPU, reduced like the headline: units of all ranks / MAX time over ranks."""
    out = {}
    for name in per_rank[0]:
        legs = [r[name] for r in per_rank]
        l0 = legs[0]
        agg = {k: v for k, v in l0.items() if not isinstance(v, (float, list, dict)) or k in ("workload",)}
        px_all = sum(l["px"] for l in legs)
        if "dt_s" in l0:                    # K steps of the boundary call, started together
            dt = max(l["dt_s"] for l in legs)
            agg.update(process_mpix_s=px_all * l0["steps"] / dt / 1e6, ms_per_step=dt / l0["steps"] * 1e3,
                       k0_avg_launch_ms=max(l["k0_ms"] for l in legs), k1_avg_launch_ms=max(l["k1_ms"] for l in legs),
                       k1_mpix_s=px_all / (max(l["k1_ms"] for l in legs) * 1e-3) / 1e6,
                       k1_ms_per_rank=[l["k1_ms"] for l in legs], px_per_gpu=l0["px"])
        if "batched_ms" in l0:              # the batched chain
            ms = max(l["batched_ms"] for l in legs)
            agg.update(batched_ms=ms, batched_ms_per_frame=ms / l0["frames"], batched_mpix_s=px_all / ms / 1e3,
                       batched_ms_per_rank=[l["batched_ms"] for l in legs],
                       frames_bit_identical_to_single_calls=all(l.get("frames_bit_identical_to_single_calls", True) for l in legs))
            for k in ("single_frame_calls_ms_per_frame", "single_frame_calls_mpix_s", "pipeline_only_batched_ms_per_frame"):
                if k in l0:
                    agg[k] = l0[k]          # rank 0's own figure (per GPU)
        agg["n_gpus"] = world
        out[name] = agg
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))            # the parent never touches a GPU
    # RCCL prints its version banner / warnings on the C-level stdout: keep the real stdout for the ONE JSON
    # line of the contract and send everything else that writes to fd 1 to stderr
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    from kinectdepthmapenhancement_amd import sharding, synth
    from kinectdepthmapenhancement_amd._native import JbfParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} != --gpus {args.gpus}")
    if args.total_frames > 0:                # the same test on every rank, before any collective: all ranks leave together
        empty = [r for r, (_, c) in enumerate(sharding.partition(args.total_frames, world)) if c < 1]
        if empty:                            # (ceil(T/N) blocks: 5 frames over 4 ranks are 2, 2, 1, 0)
            raise SystemExit(f"--total-frames {args.total_frames} in blocks of ceil(T/N) leaves rank(s) {empty} of {world} without a frame")
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    # process groups: gloo for the rendezvous, RCCL on top of it if it comes up on every rank; else the in-process
    # fallback of SURVEY 8(e) -- nothing is re-executed, the run is flagged [REPLICAS ONLY] (sharding.ShardComm)
    verdict = json.loads(os.environ["KDE_RCCL_PROBE_VERDICT"]) if os.environ.get("KDE_RCCL_PROBE_VERDICT") else None
    # (--share-device with --backend nccl: RCCL wants one device per rank, so the probes / the bring-up fail and the run goes on
    #  over gloo, flagged -- the closest a one-GPU box gets to a broken node; tests/test_gpu_sharding.py uses exactly that)
    comm = sharding.ShardComm(args.backend, local_rank, use_gpu=not args.dry_run, force_rccl_failure=args.force_rccl_failure,
                              rccl_timeout_s=args.rccl_timeout, probe=False if args.no_rccl_probe else verdict)
    barrier = comm.barrier
    use_dist = comm.active

    # ---- parameter block: every rank forms it, rank 0's is broadcast, everyone compares ----------------
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
    mine = sharding.pack_params(p, table=host_table(p))
    blk = comm.broadcast_params(mine)           # RCCL over xGMI when it is up, gloo otherwise
    p, _, _, _, table0 = sharding.unpack_params(blk)
    disagree = int(comm.allreduce_sum([0.0 if np.array_equal(mine, blk) else 1.0])[0])
    replicas_only = comm.replicas_only or disagree > 0      # flagged in the line: RCCL did not come up, or a rank's own block differs
    ranks_seen = int(round(comm.allreduce_sum([1.0])[0]))
    devices = comm.gather_objects(device_identity(rank, local_rank, args.dry_run))
    rccl = {"wanted": comm.rccl_wanted, "ok": comm.backend_used == "nccl", "backend_used": comm.backend_used, "error": comm.rccl_error,
            "probe": comm.probe, "bring_up_s": comm.bringup_s, "timeout_s": args.rccl_timeout,
            "ranks_whose_block_differs_from_rank0": disagree}

    # ---- this rank's shard of the global batch ------------------------------------------------------
    strong = args.total_frames > 0
    total_frames = args.total_frames if strong else args.frames_per_gpu * world
    first, count = sharding.partition(total_frames, world)[rank]
    first += args.first_frame
    W, H = args.width, args.height
    sharding_txt = (f"contiguous frame blocks x{world}, params broadcast from rank 0 ({comm.backend_used}"
                    + (", all ranks on cuda:0" if args.share_device else "") + ")"
                    + (f" [REPLICAS ONLY] ({comm.rccl_error or 'parameter blocks differ'})" if replicas_only else ""))
    if args.dry_run:
        barrier()
        while os.environ.get("KDE_BENCH_TEST_HANG"):        # tests/test_bench_launch.py: the parent's deadline must end us
            time.sleep(1.0)
        dt = comm.allreduce_max(1e-3 * (rank + 1))
        checksum = comm.allreduce_sum([float(first), float(count)])
        # the side legs' reductions on stand-in timings (rank r is r + 1 times slower than rank 0)
        fake = {"fhd_w19_config3": {"workload": "stand-in", "px": 32 * 1920 * 1080, "steps": 3, "dt_s": 0.03 * (rank + 1),
                                    "k0_ms": 0.5 * (rank + 1), "k1_ms": 10.0 * (rank + 1)},
                "vga_chain_batch64": {"frames": 64, "px": 64 * 640 * 480, "batched_ms": 1.7 * (rank + 1)}}
        legs = reduce_shard_legs(comm.gather_objects(fake), world)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "dry_run": True, "max_dt_over_ranks": dt, "replicas_only": replicas_only,
                              "scaling": "strong" if strong else "weak", "ranks_seen": ranks_seen, "devices": devices, "rccl": rccl,
                              "checksum": {"sum_first_frames": checksum[0], "frames": int(checksum[1])},
                              "roofline": {"fhd_w19": legs["fhd_w19_config3"]}, "also": {"vga_chain_batch64": legs["vga_chain_batch64"]},
                              "config": {"window": p.window_size, "frames_per_gpu": count, "sharding": sharding_txt}}),
                  file=real_stdout, flush=True)
        comm.close()
        return
    if count < 1:
        raise SystemExit(f"rank {rank}: empty shard ({total_frames} frames over {world} GPUs)")
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
        dt_i, st_i, _, _, _, _ = time_steps(torch, jbf, depth, color, smooth, out, args.steps, args.warmup, barrier, 0.0)
        dt_i = comm.allreduce_max(dt_i)
        idle = {"value": total_frames * W * H * args.steps / dt_i / 1e6, "ms_per_step": dt_i / args.steps * 1e3,
                "step_gpu_ms_first_median_last": [float(st_i[0]), float(np.median(st_i)), float(st_i[-1])],
                "note": "same W warm-up + K timed steps started from an idle GPU (lowest clock level), measured before the headline"}
    # ---- the headline: K x kde_jbf_process_batch --------------------------------------------------------------------
    out2 = torch.empty_like(out)            # (allocated before anything is timed: no allocation gap between the two legs)
    busy = 30.0 if args.wakeup_ms > 0 else 0.0      # --wakeup-ms 0 = the bare W + K contract, as in the from-idle leg
    dt, step_ms, _, _, woke, filler = time_steps(torch, jbf, depth, color, smooth, out, args.steps, args.warmup, barrier, args.wakeup_ms,
                                                 keep_busy_ms=busy)
    # ---- the same K steps as two separate launches, right behind it, into buffers of their own: the per-kernel times the
    # roofline object reads.  (A short wake-up of its own: the barriers between the legs idle the GPU for a moment) ----
    dt_s, _, k0_ms, k1_ms, _, _ = time_steps(torch, jbf, depth, color, smooth, out2, args.steps, args.warmup, barrier,
                                             min(args.wakeup_ms, 50.0), split=True, keep_busy_ms=busy)
    dt, dt_s = comm.allreduce_max(dt), comm.allreduce_max(dt_s)
    split_same = bool(torch.equal(out2, out))
    del out2
    smooth0 = jbf.getSmoothImage_Device(count)[0].cpu().numpy() if count > 1 else jbf.getSmoothImage_Device().cpu().numpy()
    out0 = out[0].cpu().numpy()
    checksum = comm.allreduce_sum([float(out.double().sum().item()), float(count)])
    k1_per_rank = comm.gather_objects(float(np.mean(k1_ms)))
    # ---- side legs every rank runs on its own GPU (north_star: 640x480 AND 1920x1080 batches at 1/2/4/8 GPUs) --------
    legs = None
    if not args.no_extra:
        legs = reduce_shard_legs(comm.gather_objects(shard_legs(torch, filters, synth, args, barrier)), world)

    if rank == 0:
        px_per_launch = count * W * H
        k1_avg_ms = float(np.mean(k1_ms))
        names = filters.JointBilateralFilter.variants()
        vname = names[args.variant] if args.variant >= 0 else next((nm for nm in names[1:] if nm.startswith(f"w{p.window_size}-")), None)
        entry, src = pmc_lookup(args.pmc_json, p.window_size, vname)
        roof = {"bound": "hbm",
                "frac_vga_w11": None,       # K1 on this run's batch (64 x 640x480 per GPU, window 11) / 8 TB/s
                "frac_fhd_w19": None,       # K1 on 32 x 1920x1080 per GPU, window 19: the pass north_star's 70 % target names
                "limiter": "valu-issue (K1 does 2 exp + ~30 flops per tap against 11 B/pixel; see roofline.valu)",
                "kernel": "K1 joint_bilateral_filtering",
                "measured_in": "the split leg: the same W + K steps issued as kde_jbf_presmooth_batch + kde_jbf_filter_batch right after the "
                               "headline's K boundary calls, HIP events on the launch stream"}
        roof.update(k1_roofline(px_per_launch, k1_avg_ms, entry, src))
        if (W, H, p.window_size) == (640, 480, 11):
            roof["frac_vga_w11"] = roof["frac"]
        roof["k0_avg_launch_ms"] = float(np.mean(k0_ms))
        try:        # K0 of the same profiled build: HBM traffic over its 6 B/pixel and its share of the VALU roof (profiles/pmc_bench.json)
            pj = json.load(open(args.pmc_json))
            k0e = max((k for k in pj["kernels"] if "presmooth_kernel" in k["kernel"]), key=lambda k: k["grid"]) if entry else None
            if k0e:
                roof["k0"] = {"kernel": k0e["kernel"], "avg_launch_ms": roof["k0_avg_launch_ms"], "algorithmic_bytes_per_pixel": 6.0,
                              "traffic_over_algorithmic": k0e["derived"].get("traffic_ratio"), "valu_ceiling_frac": k0e["derived"].get("valu_cycles_frac"),
                              "lds_conflict_frac": k0e["derived"].get("lds_conflict_frac"),
                              "tile_walk": "XCD bands in runs of four adjacent tiles per workgroup (inputs > 32 MiB)"}
        except Exception:       # noqa: BLE001 -- no profile: the live time stands alone
            pass
        roof["launch_ms_first_min_max"] = [float(k1_ms[0]), float(np.min(k1_ms)), float(np.max(k1_ms))]   # clock ramp shows here
        roof["launch_ms"] = {"mean": k1_avg_ms, "median": float(np.median(k1_ms)), "min": float(np.min(k1_ms)),
                             "max": float(np.max(k1_ms)), "first": float(k1_ms[0])}
        roof["k1_avg_launch_ms_per_rank"] = k1_per_rank
        value = total_frames * W * H * args.steps / dt / 1e6
        res = {
            "metric": METRIC,
            "value": value,
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "wakeup_steps_before_warmup": woke,
            "untimed_steps_enqueued_before_the_start_barrier": filler,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"JointBilateralFilter::Process (K0 pre-smooth 5/30/30 + K1) on {total_frames} x {W}x{H} "
                            f"synthetic RGB-D frames ({count} per GPU), window {p.window_size} (radius {p.window_size // 2}), "
                            f"sigma_s {p.spatial_sigma:g} px, sigma_r {p.color_sigma:g}/255, sigma_d {p.depth_sigma:g} mm",
                "timed_call": "kde_jbf_process_batch, one call per step",
                "frames_per_gpu": count, "width": W, "height": H, "window": p.window_size,
                "sharding": sharding_txt,
                "kernel_variant": (names[args.variant] if args.variant >= 0 else f"auto ({vname})"),
            },
            "ranks_seen": ranks_seen, "devices": devices, "distinct_devices": len({d.get("pci_bus_id") for d in devices}),
            "rccl": rccl, "replicas_only": replicas_only,
            "boundary_vs_split": {"step_gpu_ms_mean": float(np.mean(step_ms)), "step_gpu_ms_median": float(np.median(step_ms)),
                                  "step_gpu_ms_all": [round(float(v), 4) for v in step_ms],
                                  "split_leg_ms_per_step": dt_s / args.steps * 1e3, "split_leg_value": total_frames * W * H * args.steps / dt_s / 1e6,
                                  "value_over_split_leg_value": value / (total_frames * W * H * args.steps / dt_s / 1e6),
                                  "k0_plus_k1_ms": float(np.mean(k0_ms) + np.mean(k1_ms)), "outputs_bit_identical": split_same},
            "roofline": roof,
            "checksum": {"sum_filtered_mm": checksum[0], "frames": int(checksum[1])},
        }
        if idle:
            res["from_idle"] = idle
        if not args.no_verify:
            res["verified"] = verify_frame0(args, p, synth, first, out0, smooth0, args.variant)
        if world == 1 and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline(args, synth, args.cpu_seconds)
            res["cpu_baseline_1t"] = cpu_baseline(args, synth, max(3.0, args.cpu_seconds / 2), threads=1)
        if legs is not None:
            fhd = legs.pop("fhd_w19_config3")
            names19 = next((nm for nm in names[1:] if nm.startswith("w19-")), None)
            e19, s19 = pmc_lookup(args.pmc_json, 19, names[args.variant] if args.variant >= 0 else names19)
            fhd.update(k1_roofline(fhd["px_per_gpu"], fhd["k1_avg_launch_ms"], e19, s19), bound="hbm", limiter="valu-issue")
            res["roofline"]["fhd_w19"] = fhd
            res["roofline"]["frac_fhd_w19"] = fhd["frac"]
            res["also"] = legs
            if world == 1:
                res["also"].update(single_gpu_extras(torch, filters, synth, args))
                floor = res["also"].get("k1_w19_content_dependence", {}).get("synthetic/noelide")
                if floor:
                    fhd["k1_mpix_s_noelide"] = floor["k1_mpix_s"]
                    fhd["k1_ms_noelide"] = floor["k1_ms"]
        print(json.dumps(res), file=real_stdout, flush=True)
    barrier()           # the other ranks stay until rank 0 has checked and printed
    comm.close()


def shard_legs(torch, filters, synth, args, barrier):
    """the side legs EVERY rank runs on its own GPU, each started together (barrier) and reduced by reduce_shard_legs:
    BASELINE config 3's pass -- Process on 32 x 1920x1080 per GPU, window 19, the pass north_star's roofline target names --
    and the config-5 chain on a batch of 64 x 640x480 per GPU through the batched entry points."""
    out = {}
    w, h, n, window = 1920, 1080, 32, 19
    p = filters.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = window, 3.0, 7.65, 20.0
    color, depth = make_inputs(synth, torch, 1000, n, w, h, 2)
    smooth, res = torch.empty_like(color), torch.empty_like(depth)
    jbf = filters.JointBilateralFilter(w, h, p, max_batch=n)
    if args.variant >= 0:
        jbf.set_variant(args.variant)
    steps = max(3, min(args.steps, 10))
    busy = 40.0 if args.wakeup_ms > 0 else 0.0
    dt, _, _, _, _, _ = time_steps(torch, jbf, depth, color, smooth, res, steps, 2, barrier, min(args.wakeup_ms, 100.0),
                                   keep_busy_ms=busy)                                                   # K x the boundary call
    _, _, k0, k1, _, _ = time_steps(torch, jbf, depth, color, smooth, res, steps, 1, barrier, split=True, keep_busy_ms=busy)  # its two launches
    out["fhd_w19_config3"] = {"workload": f"JointBilateralFilter::Process (K0 + K1) on {n} x {w}x{h} per GPU, window {window}, sigma 3/7.65/20 "
                                          "(BASELINE config 3 as SURVEY 8(d) sizes it: 730 MB of algorithmic traffic per launch)",
                              "frames": n, "width": w, "height": h, "window": window, "px": n * w * h, "steps": steps, "dt_s": dt,
                              "k0_ms": float(np.mean(k0)), "k1_ms": float(np.mean(k1))}
    jbf.close()
    del color, depth, smooth, res
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_chain_batch", os.path.join(ROOT, "tools", "bench_chain_batch.py"))
    bcb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bcb)
    barrier()
    out["vga_chain_batch64"] = dict(bcb.run(torch, filters, synth, 640, 480, 64, 6), px=64 * 640 * 480)
    return out


def k1_content_legs(torch, filters, names, pre, p, depth, contents, variants, rounds=12, skip=2):
    """K1 alone on the K0-smoothed guide of each content, every kernel of `variants` ({label: variant name | None}) in
    interleaved rounds (clock / thermal drift cancels); median launch time per (content, kernel)."""
    n, h, w = depth.shape
    res = torch.empty_like(depth)
    legs = {}
    for cname, col in contents.items():
        guide = torch.empty_like(col)
        pre.presmooth_batch(col, guide)
        js = {}
        for label, vname in variants.items():
            if vname is not None and vname not in names:
                continue
            j = filters.JointBilateralFilter(w, h, p, max_batch=n)
            if vname is not None:
                j.set_variant(names.index(vname))
            js[label] = j
        times = {k: [] for k in js}
        for rnd in range(rounds):
            for k, j in js.items():
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                j.filter_batch(depth, guide, res)
                b_.record()
                torch.cuda.synchronize()
                if rnd >= skip:
                    times[k].append(a_.elapsed_time(b_))
        for k, tl in times.items():
            ms = float(np.median(tl))
            legs[f"{cname}/{k}"] = {"k1_ms": ms, "k1_mpix_s": n * w * h / ms / 1e3}
        for j in js.values():
            j.close()
        del guide
    return legs


def single_gpu_extras(torch, filters, synth, args):
    """N = 1 only, not the headline: the reference's compile-time constants on the VGA batch, the dependence of K1 on content
    (windows 11 and 19), this chip's copy ceiling, the single-frame chain of config 5, the feeder; same event timing."""
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
        dt, _, _, _, _, _ = time_steps(torch, jbf, depth, color, smooth, res, steps, 2, lambda: None)
        _, _, k0, k1, _, _ = time_steps(torch, jbf, depth, color, smooth, res, steps, 1, lambda: None, split=True)
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

    run("vga_reference_constants_w5", 640, 480, 64, 8, 5, 70.0, 50.0, 20.0)
    # ---- dependence on content (tile-level rule elision fires on smooth tiles only): K1 alone on the synthetic frames and
    # on the reference's own colour frame (input/color.jpg decode, textured), each with the default kernel and with its
    # "-noelide" twin (every tile runs the full-rule body: the data-independent floor)
    try:
        from PIL import Image
        fix = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "color_640x480.png")).convert("RGB"))[..., ::-1])
    except Exception:
        fix = None
    names = filters.JointBilateralFilter.variants()
    pre = filters.JointBilateralFilter(640, 480, max_batch=64)             # K0 with the reference's 5/30/30, as in the headline step
    p11 = filters.JointBilateralFilter.default_params()
    p11.window_size, p11.spatial_sigma, p11.color_sigma, p11.depth_sigma, p11.presmooth = 11, 3.0, 7.65, 20.0, 0
    syn_c, syn_d = make_inputs(synth, torch, 0, 64, 640, 480, 8)
    content = {"synthetic": syn_c}
    if fix is not None:
        content["reference_colour_frame_x64"] = torch.from_numpy(fix).cuda()[None].repeat(64, 1, 1, 1).contiguous()
    legs = k1_content_legs(torch, filters, names, pre, p11, syn_d, content, {"default": None, "noelide": "w11-pk2-16x16-false-v4-noelide"})
    pre.close()
    out["k1_w11_content_dependence"] = {"workload": "K1 alone on the K0-smoothed guide (as in the headline step), 64 x 640x480, window 11, sigma "
                                                    "3/7.65/20; default kernel (tile-level rule elision) and its -noelide twin (every tile runs the "
                                                    "full-rule body) in interleaved rounds, median of 10", **legs}
    del content, syn_c, syn_d
    # the pass north_star's target names: 32 x 1920x1080, window 19.  Textured content = the reference's colour frame tiled
    # 3 x 2.25 and cropped to 1920x1080; the depth frames are the synthetic ones in both legs
    pre = filters.JointBilateralFilter(1920, 1080, max_batch=32)
    p19 = filters.JointBilateralFilter.default_params()
    p19.window_size, p19.spatial_sigma, p19.color_sigma, p19.depth_sigma, p19.presmooth = 19, 3.0, 7.65, 20.0, 0
    syn_c, syn_d = make_inputs(synth, torch, 1000, 32, 1920, 1080, 2)
    content = {"synthetic": syn_c}
    if fix is not None:
        big = np.ascontiguousarray(np.tile(fix, (3, 3, 1))[:1080, :1920])
        content["reference_colour_frame_tiled_1080p_x32"] = torch.from_numpy(big).cuda()[None].repeat(32, 1, 1, 1).contiguous()
    legs = k1_content_legs(torch, filters, names, pre, p19, syn_d, content, {"default": None, "noelide": "w19-pk1-16x16-false-v4-noelide"},
                           rounds=6, skip=1)
    pre.close()
    for cname in content:
        a_, b_ = legs.get(f"{cname}/default"), legs.get(f"{cname}/noelide")
        if a_ and b_:
            legs[f"{cname}/elision_gain"] = b_["k1_ms"] / a_["k1_ms"] - 1.0
    out["k1_w19_content_dependence"] = {"workload": "K1 alone on the K0-smoothed guide, 32 x 1920x1080, window 19, sigma 3/7.65/20 (the pass north_star's "
                                                    "roofline target names); default kernel and its -noelide twin (the data-independent floor) in "
                                                    "interleaved rounds, median of 5; textured content = the reference's colour frame tiled to 1080p", **legs}
    del content, syn_c, syn_d

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
    # the same chain on a BATCH of 1080p frames (north_star's unit; the 64 x 640x480 batch is a leg every rank runs)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_chain_batch", os.path.join(ROOT, "tools", "bench_chain_batch.py"))
    bcb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bcb)
    out["fhd_chain_batch8"] = bcb.run(torch, filters, synth, 1920, 1080, 8, 6)
    db = d[None].repeat(32, 1, 1).contiguous()
    pb = torch.empty((32, H, W, 3), dtype=torch.float32, device="cuda")
    ms = timed(lambda: conv.projectiveToReal(db, pb))
    out["projectiveToReal_32xfhd"] = {"ms": ms, "GBs": 16.0 * 32 * W * H / ms / 1e6, "hbm_frac": 16.0 * 32 * W * H / ms / 1e6 / HBM_PEAK_GBS}
    return out


if __name__ == "__main__":
    main()
