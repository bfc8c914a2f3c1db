set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_baseline_configs.py tests/test_gpu_txgen.py -x -q > gpurun_out/ab_t.txt 2>&1 || { tail -40 gpurun_out/ab_t.txt; exit 1; }
tail -2 gpurun_out/ab_t.txt
run() { timeout -k 10 120 python tools/c4_run.py 8192 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$1', round(r['ms'],4), {k:round(v,4) for k,v in r['kernels_ms'].items()}, round(r['roofline']['frac'],4))
"; }
run stage; run stage
