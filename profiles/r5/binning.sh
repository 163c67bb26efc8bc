#!/bin/bash
# Round 5, experiment 2: the two-level segment binning (csrc/segment_binning.hip) -- parity of the integer path, stage times.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_binning; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_full_parity.py tests/test_hip_random_sweep.py tests/test_hip_batch.py tests/test_hip_guarded_buffers.py tests/test_hip_exact_math.py tests/test_hip_wave_reduce.py tests/test_hip_fast_alpha.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 100 2>$O/bench_err_$rep.txt | tail -1 | tee $O/bench_default_$rep.json | python -c "$P" > $O/bench_default_$rep.txt
  DGR_FAST_ALPHA=1 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | tee $O/bench_fast_$rep.json | python -c "$P" > $O/bench_fast_$rep.txt
done
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4.txt
python bench.py --no-cpu-baseline --workload config5 --steps 20 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config5.txt
python bench.py --no-cpu-baseline --variant full --workload config2 --steps 100 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config2_full.txt
cat $O/pytest.txt $O/bench_*.txt; tail -3 $O/bench_err_1.txt
