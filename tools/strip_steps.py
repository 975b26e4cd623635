"""The timed steps of bench.py and nothing after them (for a kernel trace whose last kernels are a step's):
python3 tools/strip_steps.py WIDTH HEIGHT [STEPS]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from raytracing_amd.distributed import FrameSet

W, H = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
scenes = ["example_1", "example_2", "example_3", "example_4"]
torch.cuda.set_device(0)
frames = FrameSet(scenes, W, H, device=0, rank=0, world=1, group=None, precision="mixed")
frames.prepare(seed=0)
frames.step()
frames.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    frames.step()
frames.synchronize(); torch.cuda.synchronize()
print(f"{W}x{H}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step over {steps} steps")
time.sleep(0.05)          # a gap in the trace before the last step, which the timeline shows
frames.step()
frames.synchronize(); torch.cuda.synchronize()
frames.close()
