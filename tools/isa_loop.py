"""Instruction mix of the hottest loop of one kernel in a hipcc -S listing: the innermost loop body (the last backward
branch target that contains MFMAs) — counts per opcode.  usage: python tools/isa_loop.py <file.s> <mangled-name-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if pat in n]
for name in names:
    i = s.index("\n" + name + ":")
    j = s.index("s_endpgm", i)
    lines = s[i:j].splitlines()
    labels = {l[:-1]: k for k, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    best = None
    for k, l in enumerate(lines):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            body = lines[labels[m.group(1)]:k]
            nm = sum('v_mfma' in b for b in body)
            if nm and (best is None or len(body) < len(best)):
                best = body
    if best is None:
        print(name[:80], "no MFMA loop"); continue
    ops = [b.split()[0] for b in best if b.startswith("\t") and b.split() and not b.split()[0].startswith((".", ";"))]
    c = Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_") and "mfma" not in k)
    print("%s\n  loop: %d instr, %d mfma, %d other valu, %d ds, %d salu/other" % (
        name[:100], len(ops), sum(v for k, v in c.items() if "mfma" in k), valu,
        sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if not k.startswith(("v_", "ds_")))))
    print("  ", dict(c.most_common(30)))
