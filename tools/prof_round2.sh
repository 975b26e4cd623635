# Profiles of one benchmark configuration: bash tools/prof_round2.sh CONFIG [OUTDIR-NAME]   (run on the GPU box)
# kernel stats of bench.py with the frames one after the other, FETCH_SIZE / WRITE_SIZE passes, floating-point
# instruction counters, occupancy / lane counters, and the plain bench line.  Summaries land in gpurun_out/NAME.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C=${1:-examples}
O=$GRAFT_REPO_ROOT/gpurun_out/${2:-r2_$C}
rm -rf $O; mkdir -p $O
B="python3 bench.py --config $C --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_seq -o kt -- $B --steps 3 --frames sequential > $O/bench_seq_rocprof.json 2> $O/kt_seq.err
echo kt_seq done
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c -- $B --steps 1 --warmup 0 --frames sequential > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo fetch done
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c -- $B --steps 1 --warmup 0 --frames sequential > $O/pmc_write.json 2> $O/pmc_write.err
echo write done
timeout -s KILL 280 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/pmc_flops -o c -- $B --steps 1 --warmup 0 --frames sequential > $O/pmc_flops.json 2> $O/pmc_flops.err
echo flops done
timeout -s KILL 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o c -- $B --steps 1 --warmup 0 --frames sequential > $O/pmc_sq.json 2> $O/pmc_sq.err
echo sq done
python3 tools/pmc_json.py traffic $O/pmc_fetch.json $O/pmc_traffic_$C.json $O/pmc_fetch $O/pmc_write
python3 tools/pmc_json.py flops $O/pmc_flops.json $O/pmc_flops_$C.json $O/pmc_flops $O/pmc_sq
cp $O/kt_seq/kt_kernel_stats.csv $O/kernel_stats_seq_$C.csv
find $O -name "*.csv" -size +8M -delete
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_flops $O/pmc_sq $O/kt_seq
ls $O
