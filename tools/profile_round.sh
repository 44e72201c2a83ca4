# One round's profile set for profiles/: kernel-trace statistics, HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes) and matrix-pipe counters of the default bench.py workload (greedy roll-out, B captions per step).
#   usage (on the GPU box):  bash tools/profile_round.sh r03_e [batch, default 16384 = bench.py's]
# rocprofv3 gets the program directly after `--` (no env / shell hop: the profiler has initialised the GPU by then).
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
B=${2:-16384}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras --batch $B"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS > $O/stats.log 2>&1
python3 $R/tools/h3_op_breakdown.py $O/stats > $O/h3_op_breakdown.json || true
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $O/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $O/write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $O/mfma.log 2>&1
echo mfma done
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/clk -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $O/clk.log 2>&1 || echo clk pass failed
echo clk done
python3 $R/tools/pmc_summary.py $O/fetch $O/write $B $O/mfma $O/clk $O/stats > $O/pmc_summary.json
cd $R && python3 bench.py --batch $B > $O/bench.json 2> $O/bench.err
echo bench done
