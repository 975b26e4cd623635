"""Two ranks on one GPU (gloo): the benchmark's FrameJob path with progress lines.
python -m torch.distributed.run --nproc-per-node 2 tools/debug_2rank.py [SCENE W H]"""
import faulthandler, os, sys
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("RT_DIST_BACKEND", "gloo")
from raytracing_amd.distributed import FrameJob, init_group
name = sys.argv[1] if len(sys.argv) > 1 else "example_1"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270)
rank, world, group = init_group(2)
torch.cuda.set_device(0)
job = FrameJob(name, W, H, device=0, rank=rank, world=world, group=group)
r = job.r
np.random.seed(0)
import threading, time
def watch():
    while True:
        time.sleep(10)
        print(rank, "watch: stream idle =", job.stream.query(), flush=True)
threading.Thread(target=watch, daemon=True).start()
if os.environ.get("DBG_HOST_PATH"):
    r.batch_tensor = None
with torch.cuda.stream(job.stream):
    r.choose_rows()
    print(rank, "rows", r.my_rows, flush=True)
    r0, r1 = r.my_rows
    if rank == 0:
        job.full = torch.empty((H * W, 3), dtype=torch.float64, device=job.device)
        job.tile = job.full[r0 * W:r1 * W]
    else:
        job.tile = torch.empty(((r1 - r0) * W, 3), dtype=torch.float64, device=job.device)
    r.trace_checked(job.tile.data_ptr())
    print(rank, "trace_checked", flush=True)
    r.draw_samples()
    print(rank, "samples", flush=True)
for i in range(int(os.environ.get('DBG_STEPS', '12'))):
    with torch.cuda.stream(job.stream):
        r.trace()
        print(rank, "step", i, "trace queued", flush=True)
        r.shade()
        print(rank, "step", i, "shade queued", flush=True)
        job.comm.gather_tiles(job.full, job.tile, r.bounds, W)
        print(rank, "step", i, "gathered", flush=True)
    job.synchronize()
    print(rank, "step", i, "synchronised", flush=True)
dist.barrier()
print(rank, "ok", flush=True)
