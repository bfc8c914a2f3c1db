# A/B inside one gpurun call (same box)
run() { python bench.py --no-cpu --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d['kernels_ms'].items()}, d['ber'])"; }
OFDM_FAST_UNFUSED=1 run unfused
run fused
