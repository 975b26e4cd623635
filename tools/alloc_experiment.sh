# What the first writes of freshly allocated device memory do to the step time: bash tools/alloc_experiment.sh
# (on a GPU box that has just started; the first process is the interesting one).  Runs the benchmark step with
# the treatments of DevBuf::ensure (RT_ALLOC_FILL: 0 none, 1 zero fill, 2 free + allocate again, 3 both, 4 fill
# with a bit pattern) one process after the other and samples clocks and power while each runs.
set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-alloc_experiment}
mkdir -p $O
for mode in 0 0 4 1 3 0; do
  n=$(ls $O | grep -c '^run_.*json$' || true)
  ( while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | head -c 2000; echo; sleep 0.5; done ) > $O/smi_${n}_mode$mode.txt &
  SMI=$!
  RT_ALLOC_FILL=$mode timeout -k 10 120 python3 bench.py --steps 150 --warmup 5 --no-cpu-baseline > $O/run_${n}_mode$mode.json 2> $O/run_${n}_mode$mode.err || { kill $SMI; exit 1; }
  kill $SMI
  python3 - $O/run_${n}_mode$mode.json $mode <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("mode", sys.argv[2], "ms_per_step %.3f" % d["ms_per_step"], "sequential %.3f" % d["ms_per_step_frames_sequential"], "setup_s %.2f" % d["setup_s"],
      {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
PY
done
