#!/bin/bash
# one rank through the multi-rank path (RCCL, speculative steps) against the plain single-GPU path: 4K and the strip
cd $GRAFT_REPO_ROOT; out=gpurun_out/sharded1; mkdir -p $out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RT_SHARDED_PATH=1 RT_BENCH_WATCHDOG=200
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $out/4k_1.json 2> $out/4k_1.err || exit 1
export MASTER_PORT=29518
timeout -k 10 200 python3 bench.py --gpus 1 --width 3840 --height 270 --steps 100 --warmup 5 --no-cpu-baseline > $out/strip_1.json 2> $out/strip_1.err || exit 1
echo -n "sharded path: 4K "; grep -o '"ms_per_step": [0-9.]*' $out/4k_1.json; echo -n "   strip "; grep -o '"ms_per_step": [0-9.]*' $out/strip_1.json
