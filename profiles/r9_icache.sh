#!/bin/bash
# instruction-cache and issue counters of the binning kernels on the clustered and the uniform frame
TAG=${1:-now}
OUT=$PWD/gpurun_out/r9/icache_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for scene in clustered synth-v1; do
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --views-in-flight 1 --scene $scene"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_IFETCH -d $OUT/${scene}_1 -o pmc -- $CMD > $OUT/log_${scene}_1 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS -d $OUT/${scene}_2 -o pmc -- $CMD > $OUT/log_${scene}_2 2>&1
done
cd $GRAFT_REPO_ROOT
for scene in clustered synth-v1; do
  mkdir -p $OUT/m_$scene; cp -r $OUT/${scene}_1 $OUT/m_$scene/pmc_sq1; cp -r $OUT/${scene}_2 $OUT/m_$scene/pmc_sq2
  python profiles/summarize.py $OUT/m_$scene $OUT/sum_$scene > /dev/null 2>&1
  echo "== $scene"; grep -E "kernel|bin_|render_fwd" $OUT/sum_${scene}_pmc.txt
done | tee gpurun_out/r9/icache_$TAG.txt
rm -rf $OUT/*_1 $OUT/*_2 $OUT/m_*
