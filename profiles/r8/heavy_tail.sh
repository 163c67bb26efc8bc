#!/bin/bash
# The heavy-tailed scene (dgr_amd.synth.heavy_tail_scene) at config 3's size beside synth-v1: per-stage times, ms per view, pair
# evaluations per second; then the parity tests on it.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(c["ms_per_view_one_stream"] or 0,4), "R", c["num_rendered"], "visible", c["visible"], "pair_evals/s %.3g" % (c["pair_evals_per_s"] or 0), "pairs/view", c["pair_evals_per_view"], {k: round(v*1e3,1) for k,v in c["stage_ms"].items()}, "us per 1e6 instances:", {k: round(v*1e3/(c["num_rendered"]/1e6),1) for k,v in c["stage_ms"].items()})'
for sc in synth-v1 heavy_tail; do
  python bench.py --no-cpu-baseline --steps 60 --scene $sc 2>/dev/null | tail -1 | python -c "$P" $sc
done
timeout 1500 python -m pytest tests/test_hip_heavy_tail.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6
