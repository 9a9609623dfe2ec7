set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python bench.py --save-tune /tmp/tune.json > gpurun_out/final_bench.log 2>&1
cp /tmp/tune.json gpurun_out/tune_b128.json
tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $R/bench.py --steps 3 --load-tune /tmp/tune.json --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
python $R/tools/prof_summary.py /tmp/prof_kt/*/*_kernel_trace.csv 3 > $R/gpurun_out/timed_region_v11.md
cp /tmp/prof_kt/*/*_kernel_stats.csv $R/gpurun_out/kernel_stats_full_run_v11.csv
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 $R/bench.py --steps 1 --load-tune /tmp/tune.json --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
echo fetch done
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 $R/bench.py --steps 1 --load-tune /tmp/tune.json --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1
python $R/tools/pmc_bench_summary.py /tmp/pmc_f /tmp/pmc_w > $R/gpurun_out/pmc_bench_dense_v11.md
echo all done
