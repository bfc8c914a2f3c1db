# Round-3 evidence in ONE gpurun call (same box): the driver-contract bench line (M fp32 + f64 leg + C2..C5) FIRST, on the fresh box
# (the same kernels read 10 % slower after the six PMC passes have kept the part flat out for a minute: 0.87 against 0.77 ms for the
# symbol kernel), then kernel trace stats of every leg, counters of the new C3 / C4 / C5 kernels, PMC passes + HBM traffic of the
# metric chain last.
# usage (on the GPU box): bash tools/evidence3.sh   -> gpurun_out/evidence3/ ; copy the summaries into profiles/round3/
set -o pipefail
out=gpurun_out/evidence3; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
step() { echo "== $1" >> $out/progress.txt; date >> $out/progress.txt; }
step bench
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench_20steps.json 2> $out/bench.err
t0=$SECONDS; timeout -k 10 600 python bench.py > $out/final_bench.json 2>> $out/bench.err || { echo "bench failed"; exit 1; }
echo "python bench.py (defaults: M fp32 2000 steps + cpu_baseline + f64 leg + C2..C5): $((SECONDS - t0)) s wall" > $out/bench_wall.txt
step smoke
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1 || echo "smoke failed"
step stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu --no-f64 --no-secondary --steps 600 --warmup 40 > $out/stats_bench.json 2> $out/stats.err || { echo "stats pass failed"; exit 1; }
cp $out/stats/*/*kernel_stats.csv $out/final_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats64 -- python bench.py --no-cpu --no-secondary --precision fp64 --steps 200 --warmup 20 > $out/stats_bench_f64.json 2> $out/stats64.err && cp $out/stats64/*/*kernel_stats.csv $out/f64_kernel_stats.csv
step bench2
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --force-device 0 --steps 200 --warmup 20 --frames 8192 > $out/bench_gpus2_gloo_rehearsal.json 2> $out/bench2.err || echo "2-rank rehearsal failed"
export C5_WARM_S=0.5
for c in c3 c4 c5; do
  step $c
  case $c in c3) args="4096 20";; c4) args="8192 20";; c5) args="3072 20";; esac
  timeout -k 10 300 python tools/${c}_run.py $args > $out/$c.txt 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${c}_stats -- python tools/${c}_run.py $args > /dev/null 2>&1 && cp $out/${c}_stats/*/*kernel_stats.csv $out/${c}_kernel_stats.csv
  case $c in c3) pargs="4096 3";; c4) pargs="8192 3";; c5) pargs="3072 5";; esac
  C5_WARM_S=0.02 bash tools/pmc_cmd.sh $out/${c}_pmc python tools/${c}_run.py $pargs > $out/${c}_pmc_summary.txt 2>&1
done
step c5_split
OFDM_SPLIT_NO_COOP=1 OFDM_SPLIT_NO_R2=1 timeout -k 10 300 python tools/c5_run.py 3072 20 > $out/c5_split_form.txt 2>&1
OFDM_SPLIT_NO_COOP=1 timeout -k 10 300 python tools/c5_run.py 3072 20 > $out/c5_r2_form.txt 2>&1
step sweep
timeout -k 10 300 python -m ofdm_course_amd.drivers.sweep_ber --config C5 --batches 2 --frames-per-tile 64 --json $out/sweep_c5.json > /dev/null 2>&1
step pmc
bash tools/pmc.sh $out/pmc > $out/final_pmc_summary.txt 2>&1 || echo "pmc failed"
python tools/pmc_traffic.py $out/pmc > $out/traffic.json
step clock
if [ -f ofdm-course_amd/libofdm_mi355x_diag.so ]; then bash tools/clock_probe.sh $out/clock > $out/clock_probe.txt 2>&1; fi
step ubench
tools/ubench/hbm_read > $out/hbm_read.txt 2>&1
[ -x tools/ubench/hbm_copy ] && tools/ubench/hbm_copy > $out/hbm_copy.txt 2>&1
# raw per-dispatch tables stay on the box: only the summaries travel (gpurun merges at most 64 MiB)
rm -rf $out/stats $out/stats64 $out/c5_stats $out/c3_stats $out/c4_stats $out/pmc/*/ $out/c5_pmc/*/ $out/c3_pmc/*/ $out/c4_pmc/*/ $out/clock
du -sh $out >> $out/progress.txt
step done
head -c 400 $out/final_bench.json; echo
