# A/B of the step issue on one GPU: frames joined per step (RT_FRAME_PIPELINES=0) against frames as independent pipelines.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab_pipelines}; mkdir -p $O
for mode in 0 1; do
  for cfg in "--width 3840 --height 270" "--config 1080p" ""; do
    tag=$(echo "m${mode}_$cfg" | tr -d ' -')
    RT_FRAME_PIPELINES=$mode timeout -k 10 200 python3 bench.py $cfg --no-cpu-baseline --steps 40 --warmup 3 > $O/$tag.json 2> $O/$tag.err
    python3 -c "
import json,sys
d=json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1])
print('$tag', 'ms_per_step %.3f'%d['ms_per_step'], 'seq %.3f'%d['ms_per_step_frames_sequential'], 'value %.0f'%d['value'])"
  done
done
