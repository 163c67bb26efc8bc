#!/bin/bash
# Round 5, experiment 7: the rows backward (one 4x4 block per 16-lane row) revived under the exact alpha path
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_rows; mkdir -p $O
DGR_BWD_ROWS=1 timeout 1500 python -m pytest tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_random_sweep.py tests/test_hip_batch.py tests/test_hip_guarded_buffers.py tests/test_hip_error_budget.py -q -m gpu -k "not config5" 2>&1 | grep -v amdgpu.ids | tail -12 > $O/pytest_rows.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for r in 0 1; do for rep in 1 2; do
  DGR_BWD_ROWS=$r python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "$P" > $O/bench_rows${r}_$rep.txt
done; done
DGR_BWD_ROWS=1 DGR_FAST_ALPHA=1 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "$P" > $O/bench_rows1_fast.txt
DGR_BWD_ROWS=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$P" > $O/bench_rows1_drivercmd.txt
DGR_BWD_ROWS=1 python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_rows1_config4.txt
cat $O/pytest_rows.txt; for f in $O/bench_*.txt; do echo $f; cat $f; done
