for lib in diff-gaussian-rasterization_amd/lib/libdgr_hip.so diff-gaussian-rasterization_amd/lib/exp/libdgr_full_w7.so; do
  DGR_HIP_LIB=$PWD/$lib python bench.py --variant full --workload config2 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/line.json")); st = d["config"]["stage_ms"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 4), st)
PY
done
