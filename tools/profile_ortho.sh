# Round-2 profiles of the default bench (orthoplane 1024^3 + consensus), written under gpurun_out/ (copy into profiles/):
#   1. plain run (tuner choices saved), 2. rocprofv3 kernel trace of the same command with the choices replayed,
#   3. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only, no other trace domain) over one timed pass at
#      --size 512: the same pixels per model call as at 1024^3 (128 x 512^2 = 32 x 1024^2).  At --size 1024 the
#      counter-collection run aborts inside the profiler (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT during warm-up).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py --steps 3 --save-tune /tmp/tune.json > gpurun_out/r2_bench_ortho1024.json 2> gpurun_out/r2_bench_ortho1024.log
cp /tmp/tune.json gpurun_out/r2_tune_choices_ortho1024.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $R/bench.py --steps 3 --load-tune /tmp/tune.json --no-cpu-baseline --no-forward-check > $R/gpurun_out/prof_kt.log 2>&1
python $R/tools/prof_summary.py /tmp/prof_kt/*/*_kernel_trace.csv 3 > $R/gpurun_out/r2_bench_ortho1024_timed_region.md
cp /tmp/prof_kt/*/*_kernel_stats.csv $R/gpurun_out/r2_bench_ortho1024_kernel_stats_full_run.csv
echo trace done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --size 512 --steps 1 --no-graph --load-tune /tmp/tune.json --no-cpu-baseline --no-forward-check > $R/gpurun_out/pmc_$c.log 2>&1
  echo $c done
done
python $R/tools/pmc_bench_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/gpurun_out/r2_pmc_bench_ortho512.md
echo all done
