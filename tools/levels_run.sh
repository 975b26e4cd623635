# Per-level kernel durations of one 4K frame per scene (rocprofv3 kernel trace joined with the level counts).
# Run on the GPU box: bash tools/levels_run.sh TAG SCENE...
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
for name in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/lp_$name -o lp -- python3 tools/level_profile.py $name 3840 2160 > $O/lp_$name.out 2> $O/lp_$name.err
  python3 tools/level_profile.py --join $(find $O/lp_$name -name "lp_kernel_trace.csv") gpurun_out/levels_$name.json > $O/levels_$name.txt
  cat $O/levels_$name.txt
  find $O/lp_$name -name "*.csv" -size +5M -delete
done
