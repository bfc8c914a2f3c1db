# usage: tools/pmc.sh <outdir> ; collects PMC counters for the chain kernels (separate passes)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python bench.py --no-cpu --no-f64 --no-secondary --steps 3 --warmup 1 --prewarm-seconds 0 > $out/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA
run sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python tools/pmc_sum.py $out
