#!/bin/bash
# Round 5, experiment 6: the payload-free 32-bit tile sort -- parity of the integer path everywhere, stage times
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_sort32; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_full_parity.py tests/test_hip_random_sweep.py tests/test_hip_batch.py tests/test_hip_guarded_buffers.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -12 > $O/pytest.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config3.txt
python bench.py --no-cpu-baseline --steps 100 --workload config2 --variant full 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config2_full.txt
python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4.txt
python bench.py --no-cpu-baseline --workload config5 --steps 20 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config5.txt
timeout 900 python tests/tools/soak_parity.py 150 40 23 2>&1 | grep -v amdgpu.ids | tail -3 > $O/soak_seed23.txt
cat $O/pytest.txt; for f in $O/bench_*.txt $O/soak_seed23.txt; do echo $f; cat $f; done
