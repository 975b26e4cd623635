#!/bin/bash
# The driver's multi-rank launch of bench.py, rehearsed on one GPU: N ranks over gloo, all on device 0.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/rehearse; mkdir -p $out
for n in ${RANKS:-2 4}; do
  RT_DIST_BACKEND=gloo RT_BENCH_DEVICE=0 RT_BENCH_WATCHDOG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $((29510 + n)) bench.py --gpus $n --steps 5 --warmup 1 --width ${WIDTH:-1920} --height ${HEIGHT:-1080} > $out/bench_n$n.json 2> $out/bench_n$n.err
  python3 - $n $out/bench_n$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"ranks {sys.argv[1]}: n_gpus {d['n_gpus']} step {d['ms_per_step']:.2f} ms value {d['value']:.0f} {d['unit']}; {d['config']['issue']}")
PY
done
