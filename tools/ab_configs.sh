# Same-box comparison of library builds on the strip, 1080p and 4K workloads: bash tools/ab_configs.sh NAME...   (libraries ab/NAME.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abc
for round in 1 2; do
for name in "$@"; do
  for cfg in "--width 3840 --height 270" "--config 1080p" ""; do
    tag=$(echo "${name}_$cfg" | tr -d ' -')
    RT_HIP_LIB=$GRAFT_REPO_ROOT/ab/$name.so timeout -k 10 200 python3 bench.py $cfg --no-cpu-baseline --steps 40 --warmup 3 > gpurun_out/abc/$tag.json 2> gpurun_out/abc/$tag.err || { tail -3 gpurun_out/abc/$tag.err; exit 1; }
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/abc/$tag.json') if l.startswith('{')][-1])
print('$tag'.ljust(40), 'step %.3f'%d['ms_per_step'], 'seq %.3f'%d['ms_per_step_frames_sequential'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
done
