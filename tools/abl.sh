# A/B template: run variants inside ONE gpurun call so clocks / box differences cancel.
# usage (on the GPU box): bash tools/abl.sh   -- edit the variants below; OFDM_LIB_PATH selects another build of the library.
run() { python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items()}, d['ber'])"; }
for i in 1 2; do
  OFDM_FAST_UNFUSED=1 run unfused
  run default
done
