# SQ counters of one scene's kernels: bash tools/pmc_sq.sh SCENE  (run on the GPU box)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-example_1}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$S
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o c -- python3 tools/profile_scene.py $S 3840 2160 2 > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/b -o c -- python3 tools/profile_scene.py $S 3840 2160 2 > $O/b.log 2>&1
python3 tools/pmc_summary.py $O/a $O/b
