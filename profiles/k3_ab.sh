for lib in "$@"; do
  DGR_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/line.json")); st = d["config"]["stage_ms"]
print(sys.argv[1].split("/")[-1], "K=3:", round(d["ms_per_step"], 4), "one stream:", round(d["config"]["ms_per_view_one_stream"], 4), "rf", st["render_fwd"], "rb", st["render_bwd"])
PY
done
