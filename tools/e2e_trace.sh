#!/bin/bash
# kernel trace of bench.py for one scene (its end-to-end frame comes last): bash tools/e2e_trace.sh SCENE [CONFIG]
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/e2e_${1//:/_}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py ${2:+--config $2} --scenes $1 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
