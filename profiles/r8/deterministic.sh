#!/bin/bash
# The price of dgr_set_option("deterministic_grads", 1) at config 3: bench lines with the option off / on (same box, alternating),
# then a kernel trace of the deterministic backward.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r8
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; e=c.get("grad_max_abs_err") or {}; print("deterministic_grads", sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(c["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in c["stage_ms"].items()}, "err max", e.get("max"))'
for rep in 1 2; do for m in 0 1; do
  DGR_DETERMINISTIC_GRADS=$m python bench.py --steps 100 --cpu-runs 1 2>/dev/null | tail -1 | python -c "$P" $m
done; done
cd /tmp && export TMPDIR=/tmp
DGR_DETERMINISTIC_GRADS=1 rocprofv3 --kernel-trace --stats -d /tmp/det_prof -o det -- python /root/repo/bench.py --steps 30 --views-in-flight 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/det_prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("%-70s %8s %10s" % ("kernel", "calls", "avg us"))
    for r in rows[:14]:
        print("%-70s %8s %10.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
