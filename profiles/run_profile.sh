#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [extra bench.py arguments]
# Writes gpurun_out/prof_<tag>/{stats, pmc*}; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r1}
shift || true
EXTRA="$*"
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace of the default bench command (several views in flight: kernel durations include time-sharing), and of
# the same workload one view at a time (isolated kernel durations); the counter passes use the isolated form
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/stats_default -o trace -- $BENCH > $OUT/bench_stats_default.log 2>&1
CMD="$BENCH --views-in-flight 1"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- $CMD > $OUT/bench_stats.log 2>&1
# PMC passes, one counter group per run (SQ: 8 slots; TCC: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit together)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o pmc -- $CMD > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_ATOMIC_sum -d $OUT/pmc_tcc1 -o pmc -- $CMD > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc2 -o pmc -- $CMD > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
