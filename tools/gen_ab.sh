#!/bin/bash
# generator: tests of the device stream, then kernel times of 4K example_4
set -e
out=gpurun_out/r3q; mkdir -p $out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "mt19937 or sampler" > $out/tests_gen.log 2>&1
cd /tmp; export TMPDIR=/tmp
for n in ${STREAMS:-2048}; do
export RT_MT_MAX_STREAMS=$n
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_s$n -o ex4 -- python3 $GRAFT_REPO_ROOT/bench.py --scenes example_4 --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$out/bench_s$n.json 2>$GRAFT_REPO_ROOT/$out/bench_s$n.err
done
