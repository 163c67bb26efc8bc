#!/usr/bin/env python3
"""profiles/<tag>_pmc.txt -> profiles/pmc_traffic.json: HBM bytes per launch of every stage, from the FETCH_SIZE and
WRITE_SIZE passes (KB), corrected as MI355X_MICROARCH.md (HBM) prescribes for gfx950: FETCH_SIZE tallies the 128-byte
requests of wide (16 B per lane) reads at 64 B, so it is doubled; WRITE_SIZE is calibrated here on `zero_fill_kernel`,
which writes a known 64 * P bytes and nothing else (ratio stored in the file).  usage: make_pmc_traffic.py profiles/r1_v14_final_pmc.txt
"""
import json
import re
import sys

STAGE = {"count_rank_kernel": "count_rank", "emit_instances_kernel": "emit_instances", "preprocess_bwd_kernel": "preprocess_bwd",
         "preprocess_fwd_kernel": "preprocess_fwd", "render_bwd_light_kernel": "render_bwd", "render_bwd_light_rows_kernel": "render_bwd_rows", "render_fwd_light_kernel": "render_fwd",
         "scan_blocks_kernel": "scan_blocks", "scan_tiles_kernel": "scan_tiles", "sort_tiles_kernel": "sort_tiles",
         "zero_fill_kernel": "zero_scratch", "pose_reduce_kernel": "pose_reduce", "count_lds_kernel": "count_lds",
         "scan_table_kernel": "scan_table"}
src = sys.argv[1]
vals = {}
for line in open(src):
    m = re.match(r"dgr::(\w+)(?:<[^>]*>)?\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
    if m and m.group(1) in STAGE:
        vals.setdefault(STAGE[m.group(1)], {})[m.group(2)] = float(m.group(3)) * 1024.0
P, TILES = 500000, 8160
# zero_fill_kernel runs twice per view since round 2 (the backward's accumulator rows + pose buckets, and the forward's
# padded tile counters): the per-dispatch average of the counter is compared with the average of the two known sizes
# Round 3 (the LDS count: a count_lds_kernel line is present) clears nothing in front of the forward: one launch per view.
zero_expected = (64.0 * P + 256 + 6144) if "count_lds" in vals else ((64.0 * P + 256 + 6144) + (256 + 64.0 * TILES)) / 2.0
write_cal = vals["zero_scratch"]["WRITE_SIZE"] / zero_expected
out = {"_comment": f"HBM bytes per launch at config3 (light), from {src} (one view at a time, separate FETCH_SIZE / WRITE_SIZE "
                   "passes): 2 * FETCH_SIZE + WRITE_SIZE, in bytes.  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM) "
                   "prescribes for wide reads on gfx950; WRITE_SIZE needs no correction: zero_fill_kernel writes "
                   f"{zero_expected / 1e6:.1f} MB per launch on average (launches of known size) and the counter reads {write_cal:.3f} of that.  'config3_detail' keeps the raw parts.",
       "write_calibration": write_cal, "config3": {}, "config3_detail": {}}
for k, v in sorted(vals.items()):
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out["config3"][k] = 2.0 * f + w
    out["config3_detail"][k] = {"fetch_raw": f, "write": w, "raw_sum": f + w}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["config3"], indent=1), "write calibration", write_cal)
