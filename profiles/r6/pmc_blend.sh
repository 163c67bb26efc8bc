#!/bin/bash
# HBM traffic of the blend kernels (FETCH_SIZE / WRITE_SIZE passes, one view at a time, config 3)
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r6_pmc_blend; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --views-in-flight 1 $*"
rocprofv3 --pmc FETCH_SIZE -d $O/f -o pmc -- $CMD > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o pmc -- $CMD > $O/w.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, collections
for tag in ("f", "w"):
    for db in glob.glob(f"gpurun_out/r6_pmc_blend/{tag}/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
        q = None
        for t in ("counters_collection",):
            if t in tabs:
                q = f"select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from {t} group by kernel_name, counter_name"
        if q is None:
            print("tables", tabs[:40]); continue
        for k, c, v, n in con.execute(q):
            if "render" in k or "preprocess" in k:
                print(k.split("(")[0][-60:], c, round(v / n / 1024, 1), "MB per launch" , n)
PY
