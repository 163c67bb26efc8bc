cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep -v "UserWarning\|return Variable\|assert abs\|Docs:\|Consider using" | tail -15
python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -1
DGR_SYNC_MODE=strict python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -1
python bench.py --sync-mode strict --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('strict 3 views', round(d['ms_per_step'],4), d['config']['ms_per_view_one_stream'], {k:round(v*1e3,1) for k,v in d['config']['stage_ms'].items()})"
python bench.py --sync-mode strict --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('strict one', round(d['ms_per_step'],4))"
python bench.py --sync-mode strict --workload config2 --variant full --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('config2 full strict one', round(d['ms_per_step'],4))"
