#!/bin/bash
# The first multi-GPU run, as a script (no 8-GPU node was available to any session so far: nothing below has run
# on more than one GPU).  Sweeps what decides weak-scaling efficiency for "one view per GPU + ONE fused all-reduce of the Gaussian
# gradients" -- whether RCCL picks a direct algorithm over the 7 point-to-point xGMI links, and whether the collective overlaps
# the other views' kernels:
#
#   bench.py --gpus {1,2,4,8} x --allreduce {blocking,overlap} x --blend-wgs-per-cu {0,7} x --views-in-flight {3,7}
#       at BASELINE config 3 (124 MB payload per view) and config 4 (496 MB),
#   and BASELINE config 5's pattern: --batch 4 (one all-reduce per batched step of 4 local views) and
#       --views-per-allreduce 4 (4 one-view calls per all-reduce).
#
# Checks per line: rccl_ranks == N; every rank on its own GPU; the N=1 line of a combination agrees with the plain N=1 line within
# 10 %.  Prints per line: ms per step, Mviews/s, weak-scaling efficiency against the same combination's N=1 line, and the model's
# ceilings (allreduce_model: ring / direct time at 153 GB/s per link, the collective timed alone) -- an efficiency below
# ceiling_overlapped.direct means the collective's kernels were starved or serialised; one above ceiling_serial.ring means RCCL
# did not run a ring.  Output: gpurun_out/scale_sweep/{lines.jsonl,summary.txt}.
#
#   bash profiles/scale_sweep.sh                 # on an 8-GPU node (about 25 minutes)
#   SWEEP_SHARED=1 bash profiles/scale_sweep.sh  # the same script on ONE GPU: N in {1,2}, ranks share the GPU over gloo (bench.py's
#                                                # test hooks), small step counts -- checks the script and every code path, not speed
cd "$(dirname "$0")/.."
O=gpurun_out/scale_sweep; mkdir -p $O; : > $O/lines.jsonl
if [ "$SWEEP_SHARED" = 1 ]; then
  export DGR_BENCH_SHARE_GPU=1 DGR_BENCH_BACKEND=gloo
  NS="1 2"; STEPS3=8; STEPS4=4; STEPS5=2; WARM=2; CFGS="config3"; VIFS="3"; CAPS="0 7"
else
  NS="1 2 4 8"; STEPS3=100; STEPS4=40; STEPS5=16; WARM=5; CFGS="config3 config4"; VIFS="3 7"; CAPS="0 7"
fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() {  # tag n bench-args...
  tag=$1; n=$2; shift 2
  out=$(timeout 900 python bench.py --gpus $n --no-cpu-baseline "$@" 2>$O/last.err | tail -1)
  if echo "$out" | python -c 'import sys,json; json.loads(sys.stdin.read())' 2>/dev/null; then
    echo "{\"tag\": \"$tag\", \"n\": $n, \"line\": $out}" >> $O/lines.jsonl
  else
    echo "{\"tag\": \"$tag\", \"n\": $n, \"line\": null, \"error\": \"$(tail -3 $O/last.err | tr '"\n' "' ")\"}" >> $O/lines.jsonl
  fi
}
for cfg in $CFGS; do
  steps=$STEPS3; [ $cfg = config4 ] && steps=$STEPS4
  line "$cfg plain" 1 --workload $cfg --steps $steps --warmup $WARM
  for ar in blocking overlap; do for cap in $CAPS; do for vif in $VIFS; do for n in $NS; do
    line "$cfg allreduce=$ar cap=$cap views_in_flight=$vif" $n --workload $cfg --steps $steps --warmup $WARM --allreduce $ar --blend-wgs-per-cu $cap --views-in-flight $vif
  done; done; done; done
done
if [ "$SWEEP_SHARED" = 1 ]; then W5=config2; else W5=config5; fi   # (config 5's pattern; at config 2's size on the shared GPU)
for n in $NS; do
  line "$W5 batch=4 (one all-reduce per batched step)" $n --workload $W5 --steps $STEPS5 --warmup $WARM --batch 4
  line "$W5 views_per_allreduce=4" $n --workload $W5 --steps $((STEPS5 * 4)) --warmup $WARM --views-per-allreduce 4
done
python - "$O" <<'PY' | tee $O/summary.txt
import json, sys, collections
O = sys.argv[1]
rows = [json.loads(l) for l in open(O + "/lines.jsonl")]
by = collections.defaultdict(dict)
plain = {}
bad = 0
for r in rows:
    if r["line"] is None:
        print("FAILED", r["tag"], "N =", r["n"], ":", r.get("error")); bad += 1; continue
    if r["tag"].endswith("plain"):
        plain[r["tag"].split()[0]] = r["line"]
    else:
        by[r["tag"]][r["n"]] = r["line"]
print("%-58s %2s %9s %10s %6s | %-28s %-28s alone ms" % ("combination", "N", "ms/step", "Mviews/s", "eff", "ceiling serial ring/direct", "ceiling overlapped ring/dir"))
for tag, lines in by.items():
    base = lines.get(1)
    for n in sorted(lines):
        d = lines[n]; c = d["config"]
        ranks = c.get("rccl_ranks")
        gpus = c.get("rank_gpus") or []
        notes = []
        if n > 1 and ranks != n:
            notes.append("rccl_ranks = %r" % ranks); bad += 1
        import os
        if n > 1 and os.environ.get("SWEEP_SHARED") != "1" and len({g[1] for g in gpus if isinstance(g, (list, tuple))}) != n:
            notes.append("ranks share GPUs: %r" % gpus); bad += 1
        if n == 1:
            p = plain.get(tag.split()[0])
            if p and "batch" not in tag and "views_per_allreduce" not in tag and abs(d["ms_per_step"] / p["ms_per_step"] - 1.0) > 0.10:
                notes.append("N=1 differs from the plain line by %+.0f %%" % (100 * (d["ms_per_step"] / p["ms_per_step"] - 1.0)))
        eff = (d["value"] / n) / base["value"] if base else float("nan")
        m = c.get("allreduce_model") or {}
        cs, co = m.get("ceiling_serial") or {}, m.get("ceiling_overlapped") or {}
        print("%-58s %2d %9.4f %10.6f %6.3f | %-28s %-28s %s  %s" % (
            tag, n, d["ms_per_step"], d["value"], eff,
            ("%.3f / %.3f" % (cs["ring"], cs["direct"])) if cs else "-", ("%.3f / %.3f" % (co["ring"], co["direct"])) if co else "-",
            ("%.3f" % m["measured_alone_ms"]) if m.get("measured_alone_ms") else "-", "; ".join(notes)))
print("checks failed:" if bad else "all checks passed:", bad)
PY
