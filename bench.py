"""Benchmark of the ray-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config examples|1080p|spot|many|8k] [--width .. --height ..]

Workload (--config examples, the default): the reference's example_1..4 scenes at 3840x2160 (the resolution
the headline metric is quoted on), depths (3, 5), example_4 with 30 disk
samples per hit drawn from NumPy's stream seeded with 0.  One *step* renders
all four frames: geometry of all bounce levels, lighting/shading and the
bottom-up colour pass.  Scene buffers and the disk-light samples are resident
in HBM before the timed region starts (drawing and uploading the samples is
reported separately as `setup`).

Metric: Mrays/s, rays = rays handed to the nearest-hit search (the
reference-equivalent count: sum of the batch sizes of all tree nodes whose
whole-image batch is >= 3), summed over the four frames and all GPUs.

Two timed regions of K steps each, both bracketed by barrier + synchronize:
1. the metric: the four frames are independent, each runs on its own HIP
   stream, so their latency-bound deep levels overlap; the step is issued as a
   HIP graph (one GPU: one graph per step; several GPUs: one graph per bounce
   level, the collectives between them) (`value`, `ms_per_step`);
2. the same steps with the frames one after the other and HIP events around
   every kernel launch on the kernel's stream: per-kernel durations without
   overlap, from which `kernel_ms_per_step` and the rooflines are computed
   (`ms_per_step_frames_sequential` is this region's wall time per step).

With --gpus N > 1 there is one rank (process) per GPU: either started by
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or,
when WORLD_SIZE is not set, by this script itself - the parent starts N fresh child
processes with those variables set before it has touched the GPU or torch, relays their
output, and exits non-zero if any of them does.  Every
frame is split into N contiguous row blocks of equal estimated cost; after
every bounce level one all-reduce sums the per-node batch sizes of all four
frames (the small-batch rule needs whole-image sizes); the float64 image tiles
of a step are gathered on rank 0 while the next step is computed (two output
buffers), all of them before the timed region ends.

Other configurations (same JSON line, same timed regions):
  1080p  example_1..4 at 1920x1080
  spot   a directional disk light with 64 samples per hit over a ground and a mirror ball, one frame per
         easing mode (linear, quadratic, cubic, exponential), 3840x2160 - soft-shadow heavy
  many   4096 random spheres (70 % glass) over a ground, point light, 3840x2160 - intersection bound
  8k     example_4 at 7680x4320 - the frame the multi-GPU scaling target is quoted on

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

SCENES = ["example_1", "example_2", "example_3", "example_4"]
CONFIGS = {
    "examples": (SCENES, 3840, 2160),
    "1080p": (SCENES, 1920, 1080),
    "spot": (["spot:linear", "spot:quadratic", "spot:cubic", "spot:exponential"], 3840, 2160),
    "many": (["many:4096"], 3840, 2160),
    "8k": (["example_4"], 7680, 4320),
}
FP32_VALU_PEAK_TFLOPS = 157.3   # MI355X vector FP32 with packed FMA (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (half the packed-FP32 figure)
HBM_PEAK_GBS = 8000.0
# algorithmic FLOP model: per (ray, sphere) nearest-hit test, per hit, per (hit, light sample)
# and per (shadow ray, sphere) test
F_ISECT, F_HIT, F_LIGHT, F_SHADOW = 22, 100, 50, 24


def cpu_baseline(side=360):
    """Time the NumPy float64 oracle on this host on the headline's own mix: the four example
    scenes at equal pixel counts (side x side; about 20 s of CPU work on the GPU box's host, four fifths of it
    example_4's 30 shadow rays per hit and light, which the metric does not count as rays).  The
    oracle is a restatement, not the reference: it gathers by sphere index and skips the
    reference's temporaries, and runs about three times as fast as the reference itself did on
    the development container (0.56-0.68 Mrays/s on example_1-3)."""
    from oracle import numpy_oracle
    from raytracing_amd.Scenes.scenes import build_scene
    rays, spent, per_scene = 0, 0.0, {}
    for name in SCENES:
        scene = build_scene(name, side, side)
        t0 = time.perf_counter()
        _, stats = numpy_oracle.render(scene, seed=0)
        dt = time.perf_counter() - t0
        spent += dt
        rays += stats.traced_rays()
        per_scene[name] = round(stats.traced_rays() / dt / 1e6, 4)
    threads = int(os.environ.get("OPENBLAS_NUM_THREADS", os.environ.get("OMP_NUM_THREADS", "0")) or 0)
    return {"value": rays / spent / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"NumPy float64 oracle, {'+'.join(SCENES)} at {side}x{side} each (the headline's mix of scenes), "
                      f"{rays} rays in {spent:.1f} s (NumPy elementwise code is single-threaded; BLAS threads setting "
                      f"{threads or 'default'}, host has {os.cpu_count()} logical CPUs); the oracle is about 3x as fast as "
                      f"the reference's own code (0.56-0.68 Mrays/s measured on example_1-3 on another host)",
            "per_scene": per_scene}


def measured_counters(kernel, args, world, field, pattern="pmc_traffic*.json", precision=None):
    """Per-launch mean of `field` for `kernel` from the newest committed counter measurement of this
    same workload (profiles/rNN/pmc_traffic*.json: HBM bytes, made by tools/pmc_traffic.py;
    profiles/rNN/pmc_flops*.json: executed floating-point operations, made by tools/pmc_flops.py);
    None when no measurement of this workload is committed.  `precision` is what the workload ran in
    ("mixed" when some frame's samples went through the float32 pass, else "f64")."""
    import glob
    precision = precision or args.precision
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", pattern)), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        w = rec["workload"]
        if (w["scenes"] == list(args.scenes) and w["width"] == args.width and w["height"] == args.height
                and w["precision"] == precision and w["n_gpus"] == world):
            names = [k for k in rec["kernels"] if k.split("<")[0] == kernel and field in rec["kernels"][k]]
            if names:
                n = sum(rec["kernels"][k]["launches"] for k in names)
                return sum(rec["kernels"][k][field] * rec["kernels"][k]["launches"] for k in names) / n
    return None


def spawn_ranks(n, argv):
    """Start ``n`` fresh processes of this script, one rank per GPU, and wait for them.  The parent
    never imports torch or touches the GPU.  Rank 0 prints the JSON line (the children inherit
    stdout).  If a rank fails, the others (stuck in a collective without it) are ended, and its
    exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for rank in sorted(pending):
            code = procs[rank].poll()
            if code is None:
                continue
            pending.discard(rank)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {rank} exited with code {code}; stopping the other ranks", file=sys.stderr)
        if rc != 0:
            for rank in pending:
                procs[rank].terminate()
            for rank in pending:
                try:
                    procs[rank].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[rank].kill()
            break
        time.sleep(0.05)
    return rc


def rendezvous_only(world):
    """--rendezvous-only: form the process group, sum the ranks, print a JSON line.  Exercises the
    launch path (self-started ranks or torch.distributed.run) without a GPU."""
    from raytracing_amd.distributed import init_group
    rank, world_seen, group = init_group(world)
    total = rank
    if world_seen > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([rank], dtype=torch.int64)
        if dist.get_backend(group) == "nccl":
            t = t.cuda(int(os.environ.get("LOCAL_RANK", 0)))
        dist.all_reduce(t, group=group)
        total = int(t.item())
        dist.barrier(group=group)
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world_seen, "rank_sum": total}))
    if world_seen > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="examples")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--scenes", nargs="*", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["mixed", "f64"], default="mixed")
    ap.add_argument("--frames", choices=["overlapped", "sequential"], default="overlapped",
                    help="sequential: region 1 also runs the frames one after the other (profiler runs, "
                         "so that kernel durations are free of overlap)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="only start the ranks and form the process group (no GPU work): checks the launch path")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    cfg_scenes, cfg_w, cfg_h = CONFIGS[args.config]
    custom = args.scenes is not None or args.width is not None or args.height is not None
    args.scenes = list(cfg_scenes) if args.scenes is None else args.scenes
    args.width = cfg_w if args.width is None else args.width
    args.height = cfg_h if args.height is None else args.height
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")
    if args.rendezvous_only:
        return rendezvous_only(args.gpus)

    if os.environ.get("RT_BENCH_WATCHDOG"):     # dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    import torch
    from raytracing_amd.distributed import FrameSet, init_group

    rank, world, group = init_group(args.gpus)
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if "RT_BENCH_DEVICE" in os.environ:      # rehearsal of several ranks on one GPU
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    from raytracing_amd import _native
    if local_rank >= _native.device_count():
        sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but only {_native.device_count()} are visible "
                 f"(--gpus {args.gpus})")
    torch.cuda.set_device(local_rank)
    W, H = args.width, args.height

    # ---- set-up: geometry once, then draw + upload the disk-light samples -------------------
    t_setup = time.perf_counter()
    frames = FrameSet(args.scenes, W, H, device=local_rank, rank=rank, world=world, group=group, precision=args.precision)
    frames.prepare(seed=0)       # row split, first geometry pass, samples resident in HBM, graphs captured
    jobs = frames.jobs
    setup_s = time.perf_counter() - t_setup

    def barrier():
        frames.synchronize()
        torch.cuda.synchronize()
        if group is not None:
            torch.distributed.barrier(group=group)

    def max_over_ranks(seconds):
        if group is None:
            return seconds
        on_gpu = torch.distributed.get_backend(group) == "nccl"
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
        return float(t.item())

    # ---- timed region 1: the metric.  Frames are independent, every frame has its own stream,
    # so the latency-bound deep levels of one frame overlap the others.  No per-kernel events.
    serial = args.frames == "sequential"

    def one_step():
        if serial:
            for job in jobs:
                job.step()
                job.synchronize()
        else:
            frames.step()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # ---- timed region 2: the same steps with the frames one after the other and HIP events
    # around every kernel on its stream, so that per-kernel durations are free of overlap; the
    # rooflines below come from this region
    for job in jobs:
        job.profile_enable(True)
        job.profile(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for job in jobs:
            job.step()
            job.synchronize()
    barrier()
    elapsed_seq = max_over_ranks(time.perf_counter() - t0)

    prof = {}
    prof_by_job = []
    for job in jobs:
        mine = job.profile(reset=False)
        prof_by_job.append(mine)
        for kind, rec in mine.items():
            acc = prof.setdefault(kind, {"ms": 0.0, "launches": 0})
            acc["ms"] += rec["ms"]
            acc["launches"] += rec["launches"]
        job.profile_enable(False)

    # ---- end to end, one frame at a time (single GPU only): everything a raytrace(scene) call does
    # with a warm context - geometry with the count read-back, generating and packing the disk-light
    # samples from NumPy's stream state, shading, copying the float64 image to the host
    end_to_end = None
    if group is None:
        end_to_end = {}
        host_frame = jobs[0].ctx.alloc_host((H * W, 3))      # page-locked, allocated once: the copy runs at the link's rate
        for job in jobs:
            np.random.seed(0)
            barrier()
            t0 = time.perf_counter()
            with torch.cuda.stream(job.stream):
                job.r.frame(out=host_frame)
            end_to_end[job.name] = (time.perf_counter() - t0) * 1e3
        del host_frame

    # ---- bookkeeping ----------------------------------------------------------------------
    stats = [job.stats() for job in jobs]            # global (all-rank) counts per frame
    if rank != 0:
        frames.close()
        torch.distributed.destroy_process_group()
        return

    rays = sum(s["rays"] for s in stats)
    hits = sum(s["hits"] for s in stats)
    shadow_rays = sum(s["shadow_rays"] for s in stats)
    ms_per_step = elapsed / args.steps * 1e3
    value = rays / (elapsed / args.steps) / 1e6

    # ---- rooflines (rank 0's share of the work, its own kernel times) -------------------------
    # Algorithmic work per step from the device counters:
    #   k_trace   reads link (4 B) + parent info, hit point, direction (52 B) per non-primary ray,
    #             writes info (4 B) per ray, 48 B per stored hit record, 8 B per live hit (its reflected fraction; the 8 B path
    #             length that hits entering a transparent sphere store on top are not counted) and the
    #             per-tile counters (52 B per 64 rays);   FLOPs 22 per (ray, sphere) + 100 per hit
    #   k_surface reads 52 B per shaded hit and writes 24 B; per disk-light sample 8 B (the float32 pass
    #             reads the sample's point of the unit disk as two floats) or 16 B (float64 pass: two
    #             uniforms);   FLOPs (50 + 24 * spheres) per light sample
    # Executed floating-point operations (what the instruction counters saw, per launch) come from the
    # committed counter passes of the same workload when there are any (profiles/rNN/pmc_flops*.json).
    local = [job.local_stats() for job in jobs]
    sec = {k: v["ms"] * 1e-3 / args.steps for k, v in prof.items()}
    trace_bytes = sum(56 * (s["rays"] - s["pixels"]) + (4 + 52 / 64) * s["rays"] + 48 * s["stored_records"]
                      + 8 * s["live_hits"] for s in local)
    trace_flops = sum(F_ISECT * s["rays"] * s["spheres"] + F_HIT * s["hits"] for s in local)
    mixed = all(job.precision == "mixed" for job in jobs)
    few_spheres = all(s["spheres"] <= 64 for s in local)      # the float32 sample pass and the LDS sphere loop apply
    # frames whose lighting runs in the float32 sample pass (k_surface_mixed) and the others (k_surface, float64)
    sampled = [any(k != 0 for k, _ in s["lights"]) and mixed and s["spheres"] <= 64 for s in local]
    float32_samples = any(sampled)
    ran_in = "mixed" if float32_samples else "f64"      # how the committed counter passes name the workload

    def surface_work(which):
        picked = [(s, p) for s, p, m in zip(local, prof_by_job, sampled) if m == which]
        disk = sum(s["shaded_samples"] - s["shaded_hits"] * sum(1 for k, _ in s["lights"] if k == 0)
                   for s, _ in picked if any(k != 0 for k, _ in s["lights"]))
        return {"bytes": sum(76 * s["shaded_hits"] for s, _ in picked) + (8 if which else 16) * disk,
                "flops": sum((F_LIGHT + F_SHADOW * s["spheres"]) * s["shaded_samples"] for s, _ in picked),
                "seconds": sum(p.get("surface", {"ms": 0.0})["ms"] for _, p in picked) * 1e-3 / args.steps,
                "launches": sum(p.get("surface", {"launches": 0})["launches"] for _, p in picked) / args.steps}

    def executed(kernel):
        """Executed FLOPs of one kernel per launch, split by precision, and the time the vector
        units need for them at their peaks (None without a counter pass of this workload)."""
        f32 = measured_counters(kernel, args, world, "flop_f32", "pmc_flops*.json", ran_in)
        f64 = measured_counters(kernel, args, world, "flop_f64", "pmc_flops*.json", ran_in)
        if f32 is None or f64 is None:
            return None
        return {"flop_f32_per_launch": f32, "flop_f64_per_launch": f64,
                "lanes_per_valu_instruction": measured_counters(kernel, args, world, "lanes_per_valu", "pmc_flops*.json", ran_in),
                "valu_seconds_at_peak": f32 / (FP32_VALU_PEAK_TFLOPS * 1e12) + f64 / (FP64_VALU_PEAK_TFLOPS * 1e12)}

    def roof(kind, bound, work, peak, unit, extra, kernel=None, t=None, launches=None):
        kernel = kernel or "k_" + kind
        t = sec.get(kind, 0.0) if t is None else t
        launches = (prof[kind]["launches"] / args.steps if kind in prof else 0) if launches is None else launches
        if not t or not launches:
            return None
        ex = executed(kernel)
        r = {"kernel": kernel, "bound": bound, "unit": unit, "launches_per_step": launches, "avg_launch_ms": t * 1e3 / launches,
             "traffic": measured_counters(kernel, args, world, "total", precision=ran_in)}
        if bound == "valu" and ex is not None:
            # executed operations of both precisions against the time the vector units need for them
            flops = (ex["flop_f32_per_launch"] + ex["flop_f64_per_launch"]) * launches
            r.update({"achieved": flops / t / 1e12, "peak": flops / (ex["valu_seconds_at_peak"] * launches) / 1e12,
                      "model": "executed FLOPs (instruction counters), peak = mix of the FP32 and FP64 vector peaks for this instruction mix",
                      "work_per_step": flops, "algorithmic_flops_per_step": work})
        else:
            r.update({"achieved": work / t / (1e9 if unit == "GB/s" else 1e12), "peak": peak, "work_per_step": work,
                      "algorithmic_per_launch": work / launches})
        r["frac"] = r["achieved"] / r["peak"]
        if ex is not None:
            r["executed"] = ex
            r["valu_frac"] = ex["valu_seconds_at_peak"] * launches / t
        r.update(extra)
        return r

    rooflines = {
        "trace": (roof("trace", "hbm", trace_bytes, HBM_PEAK_GBS, "GB/s",
                       {"model": "algorithmic bytes", "flops_per_step": trace_flops, "dtype": "f64"}) if few_spheres else
                  roof("trace", "valu", trace_flops, FP32_VALU_PEAK_TFLOPS, "TFLOP/s",
                       {"dtype": "f32 screening along a uniform grid over the spheres, f64 decisions",
                        "note": "without a counter pass the figure is the brute-force model of 22 FLOP per (ray, sphere), "
                                "most of which the grid never executes"})),
    }
    w32, w64 = surface_work(True), surface_work(False)
    # the float32 sample pass streams 8 bytes per sample and is bound by HBM; the float64 pass by the vector units
    rooflines["surface_mixed"] = roof("surface", "hbm", w32["bytes"], HBM_PEAK_GBS, "GB/s",
                                      {"model": "algorithmic bytes", "flops_per_step_model": w32["flops"],
                                       "dtype": "f32 samples, f64 decisions re-checked"},
                                      kernel="k_surface_mixed", t=w32["seconds"], launches=w32["launches"])
    rooflines["surface"] = roof("surface", "valu", w64["flops"], FP64_VALU_PEAK_TFLOPS, "TFLOP/s",
                                {"bytes_per_step": w64["bytes"], "dtype": "f64"},
                                kernel="k_surface", t=w64["seconds"], launches=w64["launches"])
    for r in rooflines.values():
        if r is not None and "model" not in r:
            r["model"] = "algorithmic FLOPs"
    dominant = max((k for k in rooflines if rooflines[k]),
                   key=lambda k: rooflines[k]["avg_launch_ms"] * rooflines[k]["launches_per_step"], default=None)
    roofline = rooflines.get(dominant)

    # whole-image counts of all ranks
    step_flops = sum(F_ISECT * st["spheres"] * st["rays"] + F_HIT * st["hits"]
                     + (F_LIGHT + F_SHADOW * st["spheres"]) * st["shadow_rays"] for st in stats)
    out = {
        "metric": ("Mrays/s (primary+secondary) at 3840x2160 example scenes" if args.config == "examples" and not custom else
                   f"Mrays/s (primary+secondary), {'+'.join(args.scenes)} at {W}x{H}"),
        "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "ms_per_step_frames_sequential": elapsed_seq / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{'+'.join(args.scenes)} at {W}x{H}, depths (3,5), disk-light samples from NumPy's stream "
                               f"seeded with 0 (example_4: 30 per hit, spot: 64); one step = all frames, geometry + shading",
                   "name": "custom" if custom else args.config, "scenes": list(args.scenes), "width": W, "height": H,
                   "rays_per_step": rays, "hits_per_step": hits, "shadow_rays_per_step": shadow_rays,
                   "rows_split": "contiguous cost-balanced row blocks" if frames.sharded else "none",
                   "frames": args.frames,
                   "issue": ("frames one after the other, kernel by kernel" if serial else
                             ("HIP graph" if frames.graphs is not None else "kernel by kernel") +
                             (", one all-reduce per bounce level, tiles travel during the next step" if frames.pipelined else "") +
                             (f"; geometry speculated on the previous step's batch sizes with one all-reduce per step "
                              f"({frames.speculation['steps']} steps, {frames.speculation['redone']} issued again level by level)"
                              if frames.speculate else "")),
                   "precision": ("f64 everywhere except disk-light samples: f32 with certified "
                                 "shadow decisions, f64 re-evaluation of uncertified hits") if float32_samples else "f64 everywhere"},
        "kernel_ms_per_step": {k: v["ms"] / args.steps for k, v in prof.items()},
        "setup_s": setup_s,
        "end_to_end_ms_per_frame": end_to_end,
        "roofline": roofline,
        "rooflines": rooflines,
        # the whole step against the vector peak, by the algorithmic FLOP model (22 per ray-sphere test, 100 per hit,
        # 50 + 24 * spheres per light sample): most of these operations run in float64 here (half the FP32 rate)
        "step_vs_fp32_valu_peak": {"algorithmic_flops_per_step": step_flops,
                                   "achieved_tflops": step_flops / (ms_per_step * 1e-3) / 1e12,
                                   "peak_tflops": FP32_VALU_PEAK_TFLOPS * world,
                                   # with many spheres the model counts a test per (ray, sphere) that the
                                   # grid never executes: no fraction then (rooflines hold the executed counts)
                                   "frac": (step_flops / (ms_per_step * 1e-3) / 1e12 / (FP32_VALU_PEAK_TFLOPS * world)
                                            if few_spheres else None)},
    }
    if not args.no_cpu_baseline and world == 1:      # the host baseline is timed on single-GPU runs only
        out["cpu_baseline"] = cpu_baseline()
    frames.close()
    if group is not None:
        torch.distributed.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
