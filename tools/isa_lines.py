"""Attribute the instructions of one kernel to source lines / functions.

Usage: python tools/isa_lines.py <kernel-mangled-prefix> [asm file]
The asm must come from `hipcc ... -gline-tables-only -S --cuda-device-only`.
Counts are static; the substep loop body is executed nb_substeps times per step.
"""
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
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s*$', line)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
start = next(i for i, l in enumerate(txt) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ":" in l))
counts = collections.Counter()
valu = collections.Counter()
cur = ("?", 0)
label_at = {}
order = []
for i in range(start + 1, len(txt)):
    l = txt[i].strip()
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    op = l.split()[0]
    counts[cur] += 1
    if op.startswith("v_"):
        valu[cur] += 1
    order.append((cur, op))

# aggregate by (file, function) using a function map built from the sources
import os
src_dir = "/root/repo/upkie_amd/csrc"
func_of = {}
for fname in set(f for f, _ in counts):
    p = os.path.join(src_dir, fname)
    if not os.path.exists(p):
        continue
    name = "?"
    depth = 0
    for n, line in enumerate(open(p), 1):
        m = re.match(r"^(?:template.*>\s*)?(?:UPKIE_HD|__device__|__global__|static|inline|__forceinline__|\s)*[\w:<>\*&]+\s+(\w+)\s*\(", line)
        if m and not line.startswith(" ") and not line.startswith("//"):
            name = m.group(1)
        func_of[(fname, n)] = name
agg = collections.Counter()
aggv = collections.Counter()
for k, c in counts.items():
    key = (k[0], func_of.get(k, "?"))
    agg[key] += c
    aggv[key] += valu[k]
print("total instr", sum(counts.values()), "VALU", sum(valu.values()))
for key, c in agg.most_common(40):
    print(f"{key[0]:28s} {key[1]:32s} {c:6d}  valu {aggv[key]:6d}")
if len(sys.argv) > 3:
    f = sys.argv[3]
    print("--- lines of", f)
    for k, c in sorted(counts.items()):
        if k[0] == f:
            print(k[1], c, valu[k])
