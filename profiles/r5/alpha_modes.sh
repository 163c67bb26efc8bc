#!/bin/bash
# Round 5 (driver round 3), experiment 1: which alpha path meets north_star's 1e-5, and what it costs.
# Ran at commit a227a54 (the experiment modes were removed afterwards). Needed the library built with -DDGR_ALPHA_EXPERIMENT (modes 2 = hi/lo-corrected v_exp_f32, 3 = ocml expf + IEEE division
# next to 0 = host-bit-exact exp_ref/div_ref and 1 = fast).  Output: gpurun_out/r5_alpha/.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_alpha; mkdir -p $O
./profiles/microbench/exp_variants > $O/exp_variants.txt 2>&1
timeout 900 python -m pytest tests/test_hip_wave_reduce.py tests/test_hip_exact_math.py tests/test_hip_light_parity.py tests/test_hip_full_parity.py -x -q -m gpu -k "not config5 and not config4" 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_quick.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1), "grad_err", d["config"].get("grad_max_abs_err"))'
for m in 0 1 2 3; do
  DGR_FAST_ALPHA=$m python tests/tools/error_budget.py 500000 1920 1080 2>/dev/null | tail -1 > $O/error_budget_mode${m}_config3.json
  DGR_FAST_ALPHA=$m python tests/tools/error_budget.py 100000 640 480 2>/dev/null | tail -1 > $O/error_budget_mode${m}_100k.json
  for rep in 1 2; do
    DGR_FAST_ALPHA=$m python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | tee $O/bench_mode${m}_$rep.json | python -c "$P" > $O/bench_mode${m}_$rep.txt
  done
done
python - <<PY
import json
for m in range(4):
    for tag in ("config3", "100k"):
        try: d = json.load(open("$O/error_budget_mode%d_%s.json" % (m, tag)))
        except Exception as e: print(m, tag, "failed", e); continue
        print("mode", m, tag, "int exact:", d["integer_path_exact"], "n_contrib mism:", d["n_contrib_mismatch"],
              {k[4:]: (d[k]["differing_values"], "%.1e" % d[k]["max_abs"]) for k in d if k.startswith("img_")})
        for lab in ("end_to_end", "isolated"):
            print("    ", lab, {k: "%.1e/%.1e" % (x["max_abs"], x["scale"]) for k, x in d[lab].items()})
PY
cat $O/exp_variants.txt $O/pytest_quick.txt $O/bench_mode*.txt
