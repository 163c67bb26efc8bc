for m in overlap blocking; do
DGR_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --allreduce $m 2>&1 | tail -1 > /tmp/l.json
python - <<PY
import json
try:
    d=json.load(open("/tmp/l.json")); print("$m", round(d["ms_per_step"],4), d["config"].get("gradient_allreduce"), d["config"].get("view_hbm_frac"))
except Exception as e:
    print("$m FAILED", open("/tmp/l.json").read()[-600:])
PY
done
