"""Issue-cost ledger of a blend kernel's pair loop from a gfx950 assembly listing (hipcc -S --cuda-device-only).

The blend kernels are bound by VALU issue, and on gfx950 a wave64 vector instruction does not cost the same whatever it is
(profiles/r3_valu_ilp.txt, profiles/r8/exp_poly32.txt; cycles per wave-instruction per SIMD at 8 waves per SIMD):
  fast  2.4  v_fma / v_fmac / v_fmaak / v_fmamk / v_mul / v_add / v_sub _f32, v_mov_b32, v_and / v_or / v_xor_b32, v_add / v_sub_u32,
             v_ashrrev_i32 -- with VGPR, inline-constant or 32-bit LITERAL operands
  slow  4.2  the same with an SGPR operand; every DPP form; v_cmp*, v_cndmask, v_min / v_max / v_med3, v_lshl / v_lshr / v_lshl_add,
             v_bfi, v_or3, v_cvt_*, v_rndne, v_ldexp, v_readfirstlane ...
  pk    4.65 v_pk_*_f32 (two operations)
  f64   5.2  double-pipe arithmetic (v_cvt f64<->f32 6.3)
  trans 8.2  v_rcp / v_exp / v_log / v_sqrt / v_rsq _f32, v_permlane{16,32}_swap
Usage: python profiles/asm_ledger.py file.s <substring of the kernel's mangled name> [entries per loop iteration = 2] [marker]
Prints the instructions of the innermost loop that holds `marker` (default: the wave reduction's v_permlane32_swap; the forward
blend: ds_write_b8, its contribution-tag store), by class, per list entry."""
import collections
import re
import sys

FAST = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_mul_legacy_f32"}
TRANS = {"v_rcp_f32", "v_exp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32", "v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_rcp_iflag_f32"}
COST = {"fast": 2.4, "slow": 4.2, "pk": 4.65, "f64": 5.2, "trans": 8.2}


def classify(line):
    t = line.split(";")[0].strip()
    op = t.split()[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not op.startswith("v_"):
        return None, op
    if base in TRANS:
        return "trans", base
    if base.startswith("v_pk_"):
        return "pk", base
    if "_f64" in base:
        return "f64", base
    operands = t[len(op):]
    sgpr = re.search(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|exec)(?![\w])", operands) is not None
    dpp = op.endswith("_dpp") or "row_" in operands or "quad_perm" in operands
    if base in FAST and not sgpr and not dpp:
        return "fast", base
    return "slow", base + ("(dpp)" if dpp else "(sgpr)" if sgpr and base in FAST else "")


def main():
    path, key = sys.argv[1], sys.argv[2]
    per_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    # blocks: (label line index, comment)
    def with_comment(i):  # a label's comment may continue on the following lines
        l = body[i]
        j = i + 1
        while j < len(body) and body[j].strip().startswith(";"):
            l += " " + body[j].strip()
            j += 1
        return l
    labels = [(i, with_comment(i)) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    # the innermost loop header whose body has the permlane32 swap
    marker = sys.argv[4] if len(sys.argv) > 4 else "v_permlane32_swap"
    swaps = [i for i, l in enumerate(body) if marker in l]
    assert swaps, "no " + marker + " in this kernel"
    hdr = max((i, l) for i, l in labels if "Inner Loop Header" in l and i < swaps[0])
    name = hdr[1].split(":")[0][1:]  # LBBx_y
    # loop region: every block whose comment says it belongs to this header (the latch block may precede the header)
    region = []
    cur = None
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        m2 = re.match(r"^; %bb\.\d+:(.*)", l)
        if m or m2:
            c = (m.group(2) if m else m2.group(1))
            cur = (m and m.group(1)[1:] == name) or ("Header=" + name[1:] + " ") in c + " " or ("Header=" + name[1:]) in c
            continue
        if cur and l.startswith("\t") and l.split(";")[0].strip() and not l.strip().startswith((".", ";")):
            region.append(l)
    by_class = collections.Counter()
    by_op = collections.defaultdict(collections.Counter)
    other = collections.Counter()
    for l in region:
        c, op = classify(l)
        if c is None:
            other["salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "mem"] += 1
            continue
        by_class[c] += 1
        by_op[c][op] += 1
    total = sum(COST[c] * n for c, n in by_class.items())
    print(f"{key}: loop {name}, {len(region)} instructions per iteration = {per_iter} list entries")
    for c in ("fast", "slow", "pk", "f64", "trans"):
        n = by_class[c]
        print(f"  {c:5s} {n / per_iter:6.1f} per entry x {COST[c]:4.2f} = {n * COST[c] / per_iter:6.1f} cycles   "
              + ", ".join(f"{k} {v}" for k, v in sorted(by_op[c].items(), key=lambda kv: -kv[1])))
    print(f"  VALU {sum(by_class.values()) / per_iter:.1f} instructions, {total / per_iter:.1f} issue cycles per entry;  "
          + ", ".join(f"{k} {v / per_iter:.1f}" for k, v in other.items()) + " per entry")


if __name__ == "__main__":
    main()
