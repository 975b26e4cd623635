# Same-box comparison of library builds: bash tools/ab.sh NAME... (libraries ab/NAME.so), two rounds, interleaved.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for round in 1 2; do
  for name in "$@"; do
    RT_HIP_LIB=$GRAFT_REPO_ROOT/ab/$name.so timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline ${AB_ARGS} > gpurun_out/ab/$name.$round.json 2> gpurun_out/ab/$name.$round.err || exit 1
    python3 - $name gpurun_out/ab/$name.$round.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernel_ms_per_step"]
print(f"{sys.argv[1]:>24s} step {d['ms_per_step']:.2f} seq {d['ms_per_step_frames_sequential']:.2f} | " + " ".join(f"{n} {v:.2f}" for n, v in k.items()))
PY
  done
done
