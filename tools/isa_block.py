"""Opcode sequence of the basic block of a kernel (hipcc -S listing) that holds at least N MFMAs and the most instructions of
a given kind.  usage: python tools/isa_block.py <file.s> <mangled-name-substring> [min_mfma=4] [must_contain=v_exp]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 4
must = sys.argv[4] if len(sys.argv) > 4 else "v_exp"
name = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if pat in n][0]
i = s.index("\n" + name + ":"); j = s.index(".Lfunc_end", i)
blocks, cur = [], []
for l in s[i:j].splitlines():
    if re.match(r'^\.LBB', l):
        blocks.append(cur); cur = []
    cur.append(l)
blocks.append(cur)
best = None
for b in blocks:
    nm = sum('v_mfma' in x for x in b); ne = sum(must in x for x in b)
    if nm >= nmin and ne and (best is None or ne > best[0]):
        best = (ne, b)
ops = [x.split()[0] for x in best[1] if x.startswith("\t") and x.split() and not x.split()[0].startswith((".", ";"))]
short = lambda o: re.sub(r'v_mfma_\w+', 'MFMA', o).replace("_e32", "").replace("_e64", "")
print(len(ops), "instructions")
print(" ".join(short(o) for o in ops))
m = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", s[s.index(".amdgpu_metadata"):], re.S)
print("vgpr", m.group(1) if m else None)
