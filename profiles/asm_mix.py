"""Static instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
Usage: python profiles/asm_mix.py file.s"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
cur = None
funcs = {}
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        funcs[cur] = []
        continue
    t = l.strip()
    if cur and l.startswith("\t") and t and not t.startswith((".", ";")):
        op = t.split()[0]
        funcs[cur].append(op)
        if op == "s_endpgm":
            cur = None
for name, ins in funcs.items():
    c = collections.Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith("v_"))
    s = sum(n for k, n in c.items() if k.startswith("s_"))
    ds = sum(n for k, n in c.items() if k.startswith("ds_"))
    pk = {k: n for k, n in c.items() if k.startswith("v_pk_")}
    print(name[:90], "\n   total", len(ins), "valu", v, "salu", s, "lds", ds, "packed", sum(pk.values()), pk)
