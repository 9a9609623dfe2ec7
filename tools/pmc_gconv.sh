# SQ-counter passes over the grouped 3x3 convolution kernel (tools/pmc_gconv.py); counters only with --kernel-trace
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_g1 -- python3 $R/tools/pmc_gconv.py > $R/gpurun_out/pmc_g1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc_g2 -- python3 $R/tools/pmc_gconv.py > $R/gpurun_out/pmc_g2.log 2>&1
python3 $R/tools/pmc_gconv.py --summary /tmp/pmc_g1 /tmp/pmc_g2 > $R/gpurun_out/r2_pmc_sq_gconv.md
cat $R/gpurun_out/r2_pmc_sq_gconv.md
