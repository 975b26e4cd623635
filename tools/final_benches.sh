#!/bin/bash
# the plain bench lines kept under profiles/: bash tools/final_benches.sh OUTDIR   (run on the GPU box)
set -e
out=gpurun_out/${1:-final}; mkdir -p $out
timeout -k 10 400 python3 bench.py --steps 20 --warmup 2 > $out/bench_examples.json 2> $out/bench_examples.err
echo examples done
timeout -k 10 200 python3 bench.py --config 1080p --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_1080p.json 2> $out/bench_1080p.err
timeout -k 10 200 python3 bench.py --config spot --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_spot.json 2> $out/bench_spot.err
echo spot done
timeout -k 10 300 python3 bench.py --config many --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_many.json 2> $out/bench_many.err
echo many done
timeout -k 10 300 python3 bench.py --config 8k --steps 20 --warmup 2 --no-cpu-baseline > $out/bench_8k.json 2> $out/bench_8k.err
echo 8k done
timeout -k 10 200 python3 bench.py --width 3840 --height 270 --steps 200 --warmup 5 --no-cpu-baseline > $out/bench_strip_3840x270.json 2> $out/bench_strip.err
timeout -k 10 200 python3 bench.py --config 8k --width 7680 --height 540 --steps 100 --warmup 5 --no-cpu-baseline > $out/bench_8k_strip_7680x540.json 2> $out/bench_8k_strip.err
echo strips done
