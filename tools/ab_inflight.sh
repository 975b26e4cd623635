# How much of the strip's step is latency: the same four frames twice (eight independent pipelines) against once.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab_inflight}; mkdir -p $O
S="example_1 example_2 example_3 example_4"
run() { tag=$1; shift; timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline --steps 40 --warmup 3 > $O/$tag.json 2> $O/$tag.err; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1])
print('$tag', 'ms_per_step %.3f'%d['ms_per_step'], 'seq %.3f'%d['ms_per_step_frames_sequential'], 'value %.0f'%d['value'])"; }
run once --width 3840 --height 270 --scenes $S
run twice --width 3840 --height 270 --scenes $S $S
run ex4 --width 3840 --height 270 --scenes example_4
run ex4x2 --width 3840 --height 270 --scenes example_4 example_4
run ex4x4 --width 3840 --height 270 --scenes example_4 example_4 example_4 example_4
run ex1 --width 3840 --height 270 --scenes example_1
run ex1x4 --width 3840 --height 270 --scenes example_1 example_1 example_1 example_1
RT_LEVEL_WORKGROUPS=512 run once512 --width 3840 --height 270 --scenes $S
RT_LEVEL_WORKGROUPS=256 run once256 --width 3840 --height 270 --scenes $S
