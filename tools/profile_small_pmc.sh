# L2 / HBM-side counters of the small-batch decode (the launches the document calls L2-bandwidth-bound): kernel trace,
# FETCH_SIZE and GRBM_GUI_ACTIVE + TCC_HIT/MISS passes of `bench.py --batch B`, reduced by tools/pmc_summary.py.
#   usage (on the GPU box): bash tools/profile_small_pmc.sh <tag> <batch>
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-small_pmc}
B=${2:-128}
O=$R/gpurun_out/prof_${TAG}_B$B
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--batch $B --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/mfma -- python3 $R/bench.py $ARGS > $O/mfma.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/clk -- python3 $R/bench.py $ARGS > $O/clk.log 2>&1
python3 $R/tools/pmc_summary.py $O/fetch $O/write $B $O/mfma $O/clk $O/stats > $O/pmc_summary.json
echo done
