"""Host-side cost of issuing one benchmark step (one GPU): how long the Python loop that queues
the kernels of the four frames takes, next to the GPU time of the step, and the same step
replayed from a HIP graph (captured with torch.cuda.graph on the frames' streams).

    python tools/host_issue_time.py [W H]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from raytracing_amd.distributed import FrameJob
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
    torch.cuda.set_device(0)
    jobs = []
    for name in ["example_1", "example_2", "example_3", "example_4"]:
        job = FrameJob(name, W, H)
        np.random.seed(0)
        job.prepare()
        jobs.append(job)

    def sync():
        for j in jobs:
            j.synchronize()
        torch.cuda.synchronize()

    def whole():
        for j in jobs:
            j.step()

    def by_level():
        for level in range(jobs[0].n_levels):
            for j in jobs:
                j.step_level(level)
        for j in jobs:
            j.step_finish()

    for label, fn in (("whole steps", whole), ("level by level", by_level)):
        fn(); sync()
        issue, total = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            issue.append(t1 - t0); total.append(t2 - t0)
        print(f"{W}x{H} {label}: host issue {np.median(issue)*1e3:.3f} ms, step {np.median(total)*1e3:.3f} ms", flush=True)

    # back-to-back steps (the benchmark's region)
    sync(); t0 = time.perf_counter()
    for _ in range(10):
        whole()
    sync()
    print(f"10 steps back to back: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)

    # one graph holding the four frames: fork from a capture stream into the frames' streams
    ref = [j.image().copy() for j in jobs]
    main_stream = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(main_stream):
            g.capture_begin(capture_error_mode="thread_local")
            fork = torch.cuda.Event(); fork.record(main_stream)
            for j in jobs:
                j.stream.wait_event(fork)
                j.step()
                e = torch.cuda.Event(); e.record(j.stream)
                main_stream.wait_event(e)
            g.capture_end()
    except Exception as exc:                      # noqa: BLE001
        print("graph capture failed:", repr(exc)[:300], flush=True)
        return
    g.replay(); sync()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    sync()
    print(f"graph replay: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
    same = all(np.array_equal(j.image(), r) for j, r in zip(jobs, ref))
    print("graph output identical:", same, flush=True)

    # one graph per frame, each replayed on the frame's own stream (no join between the frames)
    per_frame = []
    for j in jobs:
        gj = torch.cuda.CUDAGraph()
        with torch.cuda.stream(j.stream):
            gj.capture_begin(capture_error_mode="thread_local")
            j.ctx.trace_geometry(); j.ctx.shade()
            gj.capture_end()
        per_frame.append(gj)
    def replay_all():
        for j, gj in zip(jobs, per_frame):
            with torch.cuda.stream(j.stream):
                gj.replay()
    replay_all(); sync()
    t0 = time.perf_counter()
    for _ in range(10):
        replay_all()
    sync()
    print(f"per-frame graphs: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
    print("per-frame graph output identical:", all(np.array_equal(j.image(), r) for j, r in zip(jobs, ref)), flush=True)


if __name__ == "__main__":
    main()
