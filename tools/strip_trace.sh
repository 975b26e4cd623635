#!/bin/bash
# kernel trace of the 3840x270 strip of the four example frames, then the timeline of its last step
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/strip_trace; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/strip_steps.py 3840 270 10 > $out/steps.txt 2> $out/steps.err
cat $out/steps.txt
python3 $GRAFT_REPO_ROOT/tools/strip_timeline.py $(ls $out/prof/*kernel_trace.csv $out/prof/*/*kernel_trace.csv 2>/dev/null | head -1) 153 > $out/strip_timeline.txt
tail -12 $out/strip_timeline.txt
