#!/bin/bash
# Round 5, experiment 4: cap the blend kernels' workgroups per CU (dynamic-LDS padding) so that other views' kernels can
# co-reside; three / four views in flight, config 3.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_blend_cap; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for n in 0 7 6 5 4; do
  for k in 3 4; do
    for rep in 1 2; do
      echo "cap=$n views_in_flight=$k rep=$rep" >> $O/summary.txt
      DGR_BLEND_WGS_PER_CU=$n python bench.py --no-cpu-baseline --steps 200 --views-in-flight $k 2>/dev/null | tail -1 | python -c "$P" >> $O/summary.txt
    done
  done
done
for n in 0 6; do
  echo "driver command, cap=$n" >> $O/summary.txt
  DGR_BLEND_WGS_PER_CU=$n python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$P" >> $O/summary.txt
done
cat $O/summary.txt
