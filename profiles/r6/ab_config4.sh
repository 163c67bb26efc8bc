cd /root/repo
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for rep in 1 2; do for n in r5 default k2l5 k2l4; do
  if [ "$n" = default ]; then export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip.so; else export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip_$n.so; fi
  python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" $n
done; done
