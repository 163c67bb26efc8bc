#!/bin/bash
# End-of-round evidence (run on the GPU box): rocprofv3 kernel-trace stats + PMC passes (run_profile.sh), summaries into
# gpurun_out/prof_<tag>/, default bench line.  usage: bash profiles/final_round.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-r2}
bash profiles/run_profile.sh $TAG > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_$TAG gpurun_out/prof_$TAG/${TAG}
python bench.py > gpurun_out/bench_default_$TAG.json 2>/dev/null; tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-300
python bench.py --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_one_stream_$TAG.json
