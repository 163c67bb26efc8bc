#!/bin/bash
# Round 5, experiment 3: K1 with its rectangles in registers; EXEC-mask microbenchmark; quick parity + stage times.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_step3; mkdir -p $O
./profiles/microbench/exec_mask_f64 > $O/exec_mask_f64.txt 2>&1
timeout 1200 python -m pytest tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_batch.py tests/test_hip_guarded_buffers.py -q -m gpu -x -k "not config5" 2>&1 | grep -v amdgpu.ids | tail -8 > $O/pytest.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | tee $O/bench_default_$rep.json | python -c "$P" > $O/bench_default_$rep.txt
done
DGR_LDS_COUNT=2 python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4_seg.txt
python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4_auto.txt
DGR_LDS_COUNT=2 python bench.py --no-cpu-baseline --workload config5 --steps 20 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config5_seg.txt
python bench.py --no-cpu-baseline --workload config5 --steps 20 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config5_auto.txt
DGR_LDS_COUNT=0 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "$P" > $O/bench_default_atomics.txt
cat $O/exec_mask_f64.txt $O/pytest.txt; for f in $O/bench_*.txt; do echo $f; cat $f; done
