# usage: tools/pmc_quick.sh <outdir> [bench args...] ; the three SQ passes + GRBM only (no TCC), summary on stdout
out=$1; shift; mkdir -p $out
root=$PWD; cd /tmp && export TMPDIR=/tmp && cd $root
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python bench.py --no-cpu --steps 3 --warmup 1 $BENCH_ARGS > $out/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python tools/pmc_sum.py $out
