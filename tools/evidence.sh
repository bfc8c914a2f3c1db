# Round evidence in ONE gpurun call (same box): kernel trace stats, PMC passes, HBM traffic, then the bench line that quotes them.
# usage (on the GPU box): bash tools/evidence.sh   -> files under gpurun_out/evidence/ ; copy them into profiles/roundN/
set -o pipefail
out=gpurun_out/evidence; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu --steps 300 --warmup 40 > $out/stats_bench.json 2> $out/stats.err || { echo "stats pass failed"; exit 1; }
cp $out/stats/*/*kernel_stats.csv $out/final_kernel_stats.csv
bash tools/pmc.sh $out/pmc > $out/final_pmc_summary.txt 2>&1 || { echo "pmc failed"; exit 1; }
python tools/pmc_traffic.py $out/pmc > $out/traffic.json && cp $out/traffic.json profiles/round1/traffic.json
timeout -k 10 600 python bench.py > $out/final_bench.json 2> $out/bench.err || { echo "bench failed"; exit 1; }
head -c 300 $out/final_bench.json; echo
