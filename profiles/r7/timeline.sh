#!/bin/bash
# Round 7: GPU timeline of the driver's command (20 timed views, 21 in flight) from a rocprofv3 kernel trace.
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r7_timeline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
STEPS=${1:-20}
rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python $R/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline ${3:-} > $O/bench.log 2>&1
grep "^{\"metric\"" $O/bench.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: ms_per_step', d['ms_per_step'])"
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/profiles/r6/timeline.py $F $STEPS ${2:-20} | tee $O/timeline.txt
