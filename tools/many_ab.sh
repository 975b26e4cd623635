# Many-sphere configuration with each accelerator: bash tools/many_ab.sh [OUT]   (ACCELS="hier grid", extra environment passes through)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3many}; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many or hierarchy" --timeout 300 --timeout-method thread > $O/tests.log 2>&1; echo tests exit $?; tail -3 $O/tests.log
fi
for m in ${ACCELS:-hier grid}; do
  RT_ACCEL=$m timeout -k 10 200 python3 bench.py --config many --no-cpu-baseline --steps 5 > $O/bench_$m.json 2> $O/bench_$m.err || { tail -5 $O/bench_$m.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/bench_$m.json') if l.startswith('{')][-1])
print('accel=$m', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})"
done
