# Floating-point instruction counts, active lanes per VALU instruction and LDS bank conflicts of one
# scene's kernels: bash tools/pmc_sq_flops.sh SCENE   (run on the GPU box; prints the table)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-example_1}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_flops_$S
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/c -o c -- python3 tools/profile_scene.py $S 3840 2160 2 > $O/c.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_INT32 --output-format csv -d $O/d -o c -- python3 tools/profile_scene.py $S 3840 2160 2 > $O/d.log 2>&1
python3 tools/pmc_sq_flops.py $S $O/c $O/d
