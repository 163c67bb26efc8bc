#!/bin/bash
# one SQ counter pass for a given library: bash profiles/pmc_quick.sh <tag> [lib.so]
TAG=$1; LIB=${2:-}
OUT=$PWD/gpurun_out/pmcq_$TAG; mkdir -p $OUT
[ -n "$LIB" ] && export DGR_HIP_LIB=$PWD/$LIB
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --views-in-flight 1 $PMC_EXTRA"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc_sq1 -o pmc -- $CMD > $OUT/log1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/log2 2>&1
cd $GRAFT_REPO_ROOT; python profiles/summarize.py gpurun_out/pmcq_$TAG gpurun_out/pmcq_$TAG/sum > /dev/null 2>&1
grep -E "render_(fwd|bwd)_light" gpurun_out/pmcq_$TAG/sum_pmc.txt
