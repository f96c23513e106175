"""Basic-block listing of one kernel: instruction counts per block, the source
lines each block maps to and its branch targets (static view of the hot loop).
Usage: python tools/isa_blocks.py <mangled-prefix> [asm]"""
import collections
import re
import sys

prefix = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/isa4/g.s"
txt = open(path).read().split("\n")
files = {}
for line in txt:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', line)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
start = next(i for i, l in enumerate(txt) if l.startswith(prefix))
blocks = []  # (label, ops list, lines counter, targets)
cur = dict(label="entry", ops=[], lines=collections.Counter(), targets=[])
loc = ("?", 0)
for i in range(start + 1, len(txt)):
    l = txt[i].strip()
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = dict(label=m.group(1), ops=[], lines=collections.Counter(), targets=[])
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    op = l.split()[0]
    cur["ops"].append(op)
    cur["lines"][loc] += 1
    if op.startswith("s_cbranch") or op == "s_branch":
        cur["targets"].append((op, l.split()[-1]))
        blocks.append(cur)
        cur = dict(label=cur["label"].split("+")[0] + "+", ops=[], lines=collections.Counter(), targets=[])
blocks.append(cur)
for b in blocks:
    n = len(b["ops"])
    nv = sum(1 for o in b["ops"] if o.startswith("v_"))
    top = ", ".join(f"{f}:{ln}x{c}" for (f, ln), c in b["lines"].most_common(4))
    tg = " ".join(f"{o[10:]}->{t}" for o, t in b["targets"])
    print(f"{b['label']:12s} n={n:5d} valu={nv:5d}  [{tg}]  {top}")

if len(sys.argv) > 3:
    want = sys.argv[3].split(",")
    hist = collections.Counter()
    for b in blocks:
        if b["label"] in want:
            hist.update(b["ops"])
    print("--- op histogram of", want, "total", sum(hist.values()))
    for op, c in hist.most_common(40):
        print(f"  {op:28s} {c}")
