# rocprofv3 kernel trace of the side model PanopticBiFPN / RegNetY-6.4GF on the stack workload (256 x 512 x 512)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rg -- python3 $R/bench.py --mode stack --model bifpn_regnety --steps 3 --no-cpu-baseline --no-forward-check > $R/gpurun_out/prof_rg.log 2>&1
python $R/tools/prof_summary.py /tmp/prof_rg/*/*_kernel_trace.csv 3 > $R/gpurun_out/r2_stack256_bifpn_regnety_timed_region_v2.md
echo done
