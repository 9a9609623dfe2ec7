# Round-3 profiles, written under gpurun_out/ (copy the summaries into profiles/).  usage: bash tools/profile_r3.sh <step ...>
#   pmc1024   two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) over one timed pass of the DEFAULT bench
#             (--size 1024).  Round 2's attempt aborted with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT; with the HIP
#             runtime waiting for every kernel (AMD_SERIALIZE_KERNEL=3) the same command completes: no launch of this
#             package is malformed, the abort needs the profiler's packet rewriting under a deep asynchronous backlog
#             (--no-graph queues ~11 000 launches per plane ahead of the device).  Counters are per dispatch, so
#             serialising does not change them.
#   mitonet   bench.py --model mitonet_pr (plain run, tuner choices saved) + rocprofv3 kernel trace of the same
#             command with the choices replayed -> timed-region summary.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for step in "$@"; do
case $step in
pmc1024)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    AMD_SERIALIZE_KERNEL=3 timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc1024_$c -- \
      python3 $R/bench.py --steps 1 --warmup 1 --no-graph --load-tune $R/profiles/r2_tune_choices_ortho1024.json \
      --no-cpu-baseline --no-forward-check > $R/gpurun_out/r3_pmc1024_$c.log 2>&1 || exit 1
    echo $c done
  done
  python3 $R/tools/pmc_bench_summary.py /tmp/pmc1024_FETCH_SIZE /tmp/pmc1024_WRITE_SIZE > $R/gpurun_out/r3_pmc_bench_ortho1024.md
  ;;
mitonet)
  cd $R
  timeout -k 10 900 python bench.py --model mitonet_pr --steps 3 --cpu-size 192 --save-tune /tmp/tune_mito.json \
    > gpurun_out/r3_bench_ortho1024_mitonet_pr.json 2> gpurun_out/r3_bench_ortho1024_mitonet_pr.log || exit 1
  cp /tmp/tune_mito.json gpurun_out/r3_tune_choices_mitonet_pr.json
  tail -2 gpurun_out/r3_bench_ortho1024_mitonet_pr.log
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mito -- \
    python3 $R/bench.py --model mitonet_pr --steps 2 --load-tune /tmp/tune_mito.json --no-cpu-baseline --no-forward-check \
    > $R/gpurun_out/r3_prof_mitonet.log 2>&1 || exit 1
  python3 $R/tools/prof_summary.py /tmp/prof_mito/*/*_kernel_trace.csv 2 > $R/gpurun_out/r3_bench_ortho1024_mitonet_pr_timed_region.md
  ;;
final)
  # the round's closing evidence: default bench (tuner choices saved), rocprofv3 kernel trace of the same command with the
  # choices replayed -> timed-region summary, the post-processing in isolation, the per-slice protocol lines
  cd $R
  timeout -k 10 900 python bench.py --steps 5 --warmup 1 --save-tune gpurun_out/r3_tune_choices_ortho1024.json \
    > gpurun_out/r3_bench_ortho1024.json 2> gpurun_out/r3_bench_ortho1024.log || exit 1
  tail -1 gpurun_out/r3_bench_ortho1024.log
  cd /tmp
  timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pdl -- \
    python3 $R/bench.py --steps 2 --warmup 1 --load-tune $R/gpurun_out/r3_tune_choices_ortho1024.json --no-cpu-baseline \
    --no-forward-check > $R/gpurun_out/r3_prof_pdl.log 2>&1 || exit 1
  python3 $R/tools/prof_summary.py /tmp/prof_pdl/*/*_kernel_trace.csv 2 > $R/gpurun_out/r3_bench_ortho1024_timed_region.md
  cp /tmp/prof_pdl/*/*_kernel_stats.csv $R/gpurun_out/r3_bench_ortho1024_kernel_stats_full_run.csv
  rm -rf /tmp/prof_pdl
  cd $R
  python tools/postproc_isolated.py 1024 > gpurun_out/r3_postproc_isolated_1024.md 2>/dev/null
  for m in graph thread deferred; do python tools/bench_per_slice.py 256 $m 2>/dev/null | grep -v amdgpu.ids; done \
    > gpurun_out/r3_per_slice_protocol_final.txt
  ;;
esac
done
echo all done
