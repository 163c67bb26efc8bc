#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output (gpurun_out/prof_<tag>/...) into small text summaries for profiles/.

  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<name>

Writes <name>_kernel_stats.txt (per-kernel count / avg / min / total, the --kernel-trace --stats view) and
<name>_pmc.txt (per-kernel average of every collected counter, summed over the hardware instances that
report it: SQ counters come per shader engine, TCC counters per dispatch)."""
import glob
import os
import sqlite3
import sys


def short(name):
    name = name.replace("dgr::(anonymous namespace)::", "dgr::").replace("void ", "")
    return name.split("(")[0][:70]


def main(src, dst):
    kernel_stats(os.path.join(src, "stats_default", "trace_results.db"), dst + "_default_cmd_kernel_stats.txt")
    kernel_stats(os.path.join(src, "stats", "trace_results.db"), dst + "_kernel_stats.txt")
    pmc(src, dst)


def kernel_stats(db, path):
    out = []
    if os.path.exists(db):
        con = sqlite3.connect(db)
        rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                           "from kernels group by name order by 6 desc").fetchall()
        tot = sum(r[5] for r in rows) or 1
        out.append(f"{'kernel':72s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_us':>11s} {'%':>6s}")
        for r in rows:
            out.append(f"{short(r[0]):72s} {r[1]:6d} {r[2]/1e3:10.1f} {r[3]/1e3:10.1f} {r[4]/1e3:10.1f} {r[5]/1e3:11.1f} {100*r[5]/tot:6.2f}")
        open(path, "w").write("\n".join(out) + "\n")
        print("\n".join(out[:12]))


def pmc(src, dst):
    lines = []
    for db in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
        con = sqlite3.connect(db)
        ndisp = dict(con.execute("select name, count(distinct dispatch_id) from pmc_events group by name").fetchall())
        rows = con.execute("select name, counter_name, sum(counter_value) from pmc_events group by name, counter_name").fetchall()
        for name, ctr, total in rows:
            if "dgr::" not in name:
                continue
            lines.append(f"{short(name):60s} {ctr:24s} {total / max(ndisp[name], 1):18.1f}   per dispatch, {ndisp[name]} dispatches")
    if lines:
        hdr = "# per-dispatch totals (summed over all reporting hardware instances)\n"
        open(dst + "_pmc.txt", "w").write(hdr + "\n".join(sorted(lines)) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
