cd /root/repo
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k in ("sort_tiles","emit_instances")})'
for i in 1 2; do for a in "" _sortnodpp; do
  echo "lib$a:"; DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so python bench.py --no-cpu-baseline --steps 40 --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
done; done
