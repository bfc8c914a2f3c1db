# A/B inside one gpurun call (same box)
run() { python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items()}, d['ber'])"; }
for i in 1 2; do
OFDM_LIB_PATH=$PWD/tools/_ab/lib_a.so run prev
OFDM_PILOT_FPW=1 run new_fpw1
OFDM_PILOT_FPW=2 run new_fpw2
OFDM_PILOT_FPW=4 run new_fpw4
done
