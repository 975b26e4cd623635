#!/bin/bash
# generator: kernel times of 4K example_4 by the largest number of streams
set -e
out=gpurun_out/r3q; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for n in 2048 4096; do
export RT_MT_MAX_STREAMS=$n
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_s$n -o ex4 -- python3 $GRAFT_REPO_ROOT/bench.py --scenes example_4 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$out/bench_s$n.json 2>$GRAFT_REPO_ROOT/$out/bench_s$n.err
done
