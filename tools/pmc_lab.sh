# PMC passes over one variant of tools/h3_gemm_lab.hip (build it to tools/_lab/lab2 first); usage: bash tools/pmc_lab.sh <variant> <shape>
# (the TA_* / SQ_INST_LEVEL_* counters hung a pass for 5 minutes on this pool: left out)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export LAB_ONLY=${1:-10} LAB_SHAPE=${2:-1} LAB_IT=4
BIN=${3:-lab2}
rm -rf $R/gpurun_out/pmc_lab; mkdir -p $R/gpurun_out/pmc_lab
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_lab/p$i -- $R/tools/_lab/$BIN > $R/gpurun_out/pmc_lab/p$i.log 2>&1 || echo pass $i failed
done
echo done
