#!/bin/bash
# A/B of experimental builds of the library: bash profiles/lib_sweep.sh <lib.so> [<lib.so> ...]   (config 3)
for lib in "$@"; do
  DGR_HIP_LIB=$PWD/$lib python bench.py --workload config3 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/line.json")); st = d["config"]["stage_ms"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 4), {k: st[k] for k in ("render_fwd", "render_bwd", "preprocess_bwd", "preprocess_fwd")})
PY
done
