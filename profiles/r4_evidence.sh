#!/bin/bash
# round-4 evidence: rocprofv3 passes of the default command (run_profile.sh), the kernel trace of a batched step, bench lines
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash profiles/final_round.sh r4
OUT=$PWD/gpurun_out/prof_r4
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_batch4 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 4 > $OUT/bench_stats_batch4.log 2>&1 )
python - <<'PY'
import sys, os
sys.path.insert(0, "profiles")
import summarize
summarize.kernel_stats("gpurun_out/prof_r4/stats_batch4/trace_results.db", "gpurun_out/prof_r4/r4_batch4_kernel_stats.txt")
PY
python bench.py --no-cpu-baseline --batch 4 --steps 60 2>/dev/null | tail -1 > gpurun_out/bench_batch4_r4.json
python bench.py --no-cpu-baseline --batch 8 --steps 40 2>/dev/null | tail -1 > gpurun_out/bench_batch8_r4.json
ls gpurun_out/prof_r4 | head -30
