# Round-2 evidence in ONE gpurun call (same box): kernel trace stats + PMC passes + HBM traffic of bench.py, the bench line that
# quotes them, the self-launched 2-rank rehearsal, and stats / counters of the secondary configurations (C3, C4, C5, Task5_part2).
# usage (on the GPU box): bash tools/evidence2.sh   -> gpurun_out/evidence2/ ; copy into profiles/round2/
set -o pipefail
out=gpurun_out/evidence2; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
step() { echo "== $1" >> $out/progress.txt; date >> $out/progress.txt; }
step pmc
bash tools/pmc.sh $out/pmc > $out/final_pmc_summary.txt 2>&1 || { echo "pmc failed"; exit 1; }
python tools/pmc_traffic.py $out/pmc > $out/traffic.json && mkdir -p profiles/round2 && cp $out/traffic.json profiles/round2/traffic.json
step bench
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_20steps.json 2> $out/bench.err
timeout -k 10 600 python bench.py > $out/final_bench.json 2>> $out/bench.err || { echo "bench failed"; exit 1; }
step stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu --steps 600 --warmup 40 > $out/stats_bench.json 2> $out/stats.err || { echo "stats pass failed"; exit 1; }
cp $out/stats/*/*kernel_stats.csv $out/final_kernel_stats.csv
step bench2
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 200 --warmup 20 --frames 8192 > $out/bench_gpus2_gloo_rehearsal.json 2> $out/bench2.err || echo "2-rank rehearsal failed"
step clock
bash tools/clock_probe.sh $out/clock > $out/clock_probe.txt 2>&1
export C5_WARM_S=0.5
step c5
timeout -k 10 300 python tools/c5_run.py 3072 20 > $out/c5.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5_stats -- python tools/c5_run.py 3072 20 > /dev/null 2>&1 && cp $out/c5_stats/*/*kernel_stats.csv $out/c5_kernel_stats.csv
C5_WARM_S=0.02 bash tools/pmc_cmd.sh $out/c5_pmc python tools/c5_run.py 3072 5 > $out/c5_pmc_summary.txt 2>&1
timeout -k 10 300 python -m ofdm_course_amd.drivers.sweep_ber --config C5 --batches 2 --frames-per-tile 64 --json $out/sweep_c5.json > /dev/null 2>&1
step c3
timeout -k 10 300 python tools/c3_run.py 4096 10 > $out/c3.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_stats -- python tools/c3_run.py 4096 10 > /dev/null 2>&1 && cp $out/c3_stats/*/*kernel_stats.csv $out/c3_kernel_stats.csv
step part2
timeout -k 10 400 python tools/part2_run.py --precision fp32 > $out/part2.txt 2>&1
timeout -k 10 300 python tools/part2_run.py --random --precision fp32 > $out/part2_random.txt 2>&1
bash tools/pmc_cmd.sh $out/part2_pmc python tools/part2_run.py --random --precision fp32 --no-percall --reps 1 > $out/part2_pmc_summary.txt 2>&1
step configs
timeout -k 10 600 python tools/bench_configs.py > $out/secondary_configs.jsonl 2> $out/configs.err
bash tools/pmc_cmd.sh $out/c4_pmc python tools/bench_configs.py C4 > $out/c4_pmc_summary.txt 2>&1
step ubench
tools/ubench/hbm_read > $out/hbm_read.txt 2>&1
# raw per-dispatch tables stay on the box: only the summaries travel (gpurun merges at most 64 MiB)
rm -rf $out/stats $out/c5_stats $out/c3_stats $out/pmc/*/ $out/c5_pmc/*/ $out/part2_pmc/*/ $out/c4_pmc/*/ $out/clock
du -sh $out >> $out/progress.txt
step done
head -c 400 $out/final_bench.json; echo
