# usage: tools/pmc_cmd.sh <outdir> <program and args...> ; SQ passes (incl. MFMA counters) + GRBM + TCC of an arbitrary command,
# per-kernel summary on stdout.  The program itself follows `--` of rocprofv3 (no shell / env wrapper in between).
out=$1; shift; mkdir -p $out
root=$PWD; cd /tmp && export TMPDIR=/tmp && cd $root
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC_SET --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || echo "pass $name failed"; }
PMC_SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" run sq1 "$@"
PMC_SET="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA" run sq2 "$@"
PMC_SET="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_CYCLES SQ_BUSY_CU_CYCLES" run sq3 "$@"
PMC_SET="GRBM_GUI_ACTIVE GRBM_COUNT" run grbm "$@"
PMC_SET="FETCH_SIZE" run tcc1 "$@"
PMC_SET="WRITE_SIZE" run tcc2 "$@"
python tools/pmc_sum.py $out
