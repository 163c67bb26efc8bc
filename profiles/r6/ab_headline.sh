cd /root/repo
for rep in 1 2 3 4; do for n in default ntld; do
  export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip$( [ $n = default ] || echo _$n ).so
  for K in 20 200; do python bench.py --no-cpu-baseline --steps $K --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n steps=$K', round(d['ms_per_step'],4))"; done
done; done
