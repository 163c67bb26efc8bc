#!/bin/bash
# The lines whose default changed with 21 views in flight (the rest of profiles/r7_bench_* stands).
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_evidence2; mkdir -p $O
B="python bench.py"
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd.json
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd_2.json
$B 2>/dev/null | tail -1 > $O/bench_config3_light.json
$B --tight-cull --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tight_cull.json
$B --scene clustered --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_clustered.json
$B --workload config2 --variant full 2>/dev/null | tail -1 > $O/bench_config2_full.json
$B --workload config4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4_light_view.json
$B --workload config5 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config5_light_view.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob
db=glob.glob("gpurun_out/r7_evidence2/stats_default/*.db")[0]
con=sqlite3.connect(db)
rows=con.execute("select name,count(*),avg(end-start),min(end-start),max(end-start),sum(end-start) from kernels group by name order by 6 desc").fetchall()
tot=sum(r[5] for r in rows)
out=[f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_us':>11s} {'%':>6s}"]
for r in rows[:14]:
    n=r[0].replace("dgr::(anonymous namespace)::","dgr::").replace("void ","").split("(")[0][:70]
    out.append(f"{n:72s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {r[5]/1e3:11.1f} {100*r[5]/tot:6.2f}")
open("gpurun_out/r7_evidence2/r7_default_cmd_kernel_stats.txt","w").write("\n".join(out)+"\n")
PY
rm -rf $O/stats_default
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split('/')[-1][6:-5], "ms/step", round(d["ms_per_step"],4), "one", c.get("ms_per_view_one_stream") and round(c["ms_per_view_one_stream"],4), "K", c["views_in_flight"], "frac", round(r["frac"],4), "graph", c.get("ms_per_step_hipgraph_replay"), "err", c.get("grad_max_abs_err",{}).get("max"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
cat $O/summary.txt; head -8 $O/r7_default_cmd_kernel_stats.txt
