# Copy the summaries of tools/prof_round.sh from gpurun_out/ into profiles/: bash tools/collect_profiles.sh r1c r1 c
set -e
SRC=gpurun_out/$1; DST=profiles/$2; P=$3
mkdir -p $DST
cp $SRC/kt_seq/kt_kernel_stats.csv $DST/${P}_frames_sequential_kernel_stats.csv
cp $SRC/kt_ovl/kt_kernel_stats.csv $DST/${P}_frames_overlapped_kernel_stats.csv
cp $SRC/bench_seq_rocprof.json $DST/${P}_frames_sequential_bench_under_rocprof.json
cp $SRC/bench_ovl_rocprof.json $DST/${P}_frames_overlapped_bench_under_rocprof.json
cp $SRC/pmc_summary.txt $DST/${P}_pmc_fetch_write_summary.txt
cp $SRC/pmc_traffic.json $DST/pmc_traffic.json
[ -f $SRC/kt_many/kt_kernel_stats.csv ] && cp $SRC/kt_many/kt_kernel_stats.csv $DST/${P}_config4_4096_spheres_kernel_stats.csv
[ -f $SRC/config4_many.json ] && cp $SRC/config4_many.json $DST/${P}_config4_4096_spheres.json
[ -f $SRC/bench.json ] && cp $SRC/bench.json $DST/${P}_bench.json
true
