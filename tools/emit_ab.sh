# Generator kernel time of 4K example_4 with different builds: bash tools/emit_ab.sh NAME...   (libraries ab/NAME.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/emit_ab
for name in "$@"; do
  rm -rf gpurun_out/emit_ab/$name
  RT_HIP_LIB=$GRAFT_REPO_ROOT/ab/$name.so timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/emit_ab/$name -o kt -- python3 bench.py --scenes example_4 --no-cpu-baseline --steps 1 > gpurun_out/emit_ab/$name.json 2> gpurun_out/emit_ab/$name.err || exit 1
  echo $name $(grep -E "k_mt_emit" gpurun_out/emit_ab/$name/kt_kernel_stats.csv | sed "s/(.*)\"//" | cut -d, -f4)
  rm -rf gpurun_out/emit_ab/$name
done
