#!/bin/bash
# A/B of two builds of the library through the compiled binding: bash profiles/r9_ab_libs.sh <dirA> <dirB> <out.txt> [reps]
# (dirs under diff-gaussian-rasterization_amd/, each holding a libdgr_hip.so; the file in lib/ is swapped and B is left in place)
# HBM counters of the four big kernels per launch (separate FETCH_SIZE / WRITE_SIZE passes), then stage times alternating.
cd "$(dirname "$0")/.."
A=$1; B=$2; TXT=gpurun_out/r9/$3; REPS=${4:-3}
mkdir -p gpurun_out/r9; rm -f $TXT
R=$PWD; PKG=diff-gaussian-rasterization_amd
cd /tmp && export TMPDIR=/tmp
for v in $A $B; do
  cp $R/$PKG/$v/libdgr_hip.so $R/$PKG/lib/libdgr_hip.so
  OUT=$R/gpurun_out/r9/ab_pmc_$v; rm -rf $OUT; mkdir -p $OUT
  for pass in "FETCH_SIZE TCC_EA0_RDREQ_sum TCC_ATOMIC_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    timeout 150 rocprofv3 --pmc $pass -d $OUT/pmc_${pass%% *} -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --views-in-flight 1 > $OUT/log 2>&1
  done
  (cd $R && python profiles/summarize.py gpurun_out/r9/ab_pmc_$v gpurun_out/r9/ab_pmc_$v/sum > /dev/null 2>&1; echo "== $v"; grep -E "render_(fwd|bwd)_light|preprocess_(fwd|bwd)" gpurun_out/r9/ab_pmc_$v/sum_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE") | tee -a $R/$TXT
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
done
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), "strict", round(d["config"].get("ms_per_view_strict_one_stream") or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render") or k.startswith("pre")})'
run() { python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in $(seq $REPS); do for v in $A $B; do
cp $PKG/$v/libdgr_hip.so $PKG/lib/libdgr_hip.so
echo -n "$v mapping  : "; run
echo -n "$v tracking : "; run --tracking
done; done 2>&1 | tee -a $TXT
cp $PKG/$B/libdgr_hip.so $PKG/lib/libdgr_hip.so
