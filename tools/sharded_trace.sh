#!/bin/bash
cd /tmp; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/sharded1; mkdir -p $out
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RT_SHARDED_PATH=1 RT_BENCH_WATCHDOG=200
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --width 3840 --height 270 --steps 30 --warmup 5 --no-cpu-baseline > $out/strip_t.json 2> $out/strip_t.err
