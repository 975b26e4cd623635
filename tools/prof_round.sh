# Profiles of one round: kernel stats of bench.py (frames sequential and overlapped), counter calibration,
# FETCH_SIZE / WRITE_SIZE passes.  Run on the GPU box: bash tools/prof_round.sh ; copy summaries to profiles/rNN/.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r1e}
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_seq -o kt -- python3 bench.py --steps 3 --no-cpu-baseline --frames sequential > $O/bench_seq_rocprof.json 2> $O/kt_seq.err
echo kt_seq done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ovl -o kt -- python3 bench.py --steps 3 --no-cpu-baseline > $O/bench_ovl_rocprof.json 2> $O/kt_ovl.err
echo kt_ovl done
[ -x tools/pmc_calib ] || hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -o c -- tools/pmc_calib > $O/cal_fetch.log 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -o c -- tools/pmc_calib > $O/cal_write.log 2>&1
echo calib done
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames sequential > $O/pmc_fetch.json 2> $O/pmc_fetch.err
echo fetch done
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --frames sequential > $O/pmc_write.json 2> $O/pmc_write.err
echo write done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_many -o kt -- python3 tools/config_runs.py many > $O/config4_many_rocprof.json 2> $O/kt_many.err
echo many done
timeout -k 10 300 python3 tools/config_runs.py many > $O/config4_many.json 2> $O/config4_many.err
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
echo bench done
python3 tools/pmc_summary.py $O/cal_fetch $O/cal_write $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json
find $O -name "*.csv" -size +20M -delete
ls -R $O | head -50
