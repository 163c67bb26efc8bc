cd /root/repo
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render")})'
for rep in 1 2; do for a in default fabl1 fabl2 fabl3; do
  if [ $a = default ]; then L=libdgr_hip.so; else L=libdgr_hip_$a.so; fi
  DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/$L python bench.py --no-cpu-baseline --steps 20 --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P" $a
done; done
