# rocprofv3 kernel trace of the default bench (orthoplane 1024^3), timed region summarised by tools/prof_summary.py
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py --steps 2 --save-tune /tmp/tune.json --no-cpu-baseline --no-forward-check > gpurun_out/prof_pre.json 2> gpurun_out/prof_pre.log
cp /tmp/tune.json gpurun_out/tune_ortho1024.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $R/bench.py --steps 2 --load-tune /tmp/tune.json --no-cpu-baseline --no-forward-check > $R/gpurun_out/prof_kt.log 2>&1
python $R/tools/prof_summary.py /tmp/prof_kt/*/*_kernel_trace.csv 2 > $R/gpurun_out/ortho1024_timed_region.md
cp /tmp/prof_kt/*/*_kernel_stats.csv $R/gpurun_out/ortho1024_kernel_stats_full_run.csv
echo done
