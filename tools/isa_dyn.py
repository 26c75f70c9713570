#!/usr/bin/env python3
"""Static estimate of the dynamic instruction mix of one kernel's outermost loop body from csrc/pgas_api.s (make asm).
usage: isa_dyn.py MANGLED_PREFIX trip1,trip2,...   (trip counts of the inner loops in order of appearance; default 10)
Straight-line code outside any inner loop counts once; conditional blocks count as executed (upper bound)."""
import re
import sys
from collections import Counter

asm = open("bayesian-inference-with-explicit-and-implicit-prior-knowledge_amd/csrc/pgas_api.s").read().split("\n")
pref = sys.argv[1]
trips = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else []
start = next(i for i, l in enumerate(asm) if l.startswith(pref) and ":" in l[:len(pref) + 120] and not l.startswith("\t"))
end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
lines = asm[start:end]
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
# inner loops = labels that are targets of a backward branch
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i))
loops.sort()
# keep innermost, non-nested; the outermost (first label spanning everything) is the time loop
outer = [lp for lp in loops if any(o[0] > lp[0] and o[1] < lp[1] for o in loops)]
inner = [lp for lp in loops if lp not in outer]
body_lo = min((o[0] for o in outer), default=0)
body_hi = max((o[1] for o in outer), default=len(lines) - 1)


def count(lo, hi):
    c = Counter()
    for l in lines[lo:hi + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c[t.split()[0]] += 1
    return c


tot = Counter()
prev = body_lo
k = 0
for lo, hi in inner:
    if lo < body_lo or hi > body_hi:
        continue
    for op, v in count(prev, lo - 1).items():
        tot[op] += v
    t = trips[k] if k < len(trips) else 10
    c = count(lo, hi)
    print(f"inner loop {k}: lines {lo}-{hi} insts {sum(c.values())} valu {sum(v for o, v in c.items() if o.startswith('v_'))} x trip {t}")
    for op, v in c.items():
        tot[op] += v * t
    prev = hi + 1
    k += 1
for op, v in count(prev, body_hi).items():
    tot[op] += v
valu = sum(v for o, v in tot.items() if o.startswith("v_"))
print(f"per thread and outer iteration: total {sum(tot.values())} valu {valu} salu {sum(v for o, v in tot.items() if o.startswith('s_'))}")
grp = Counter()
for o, v in tot.items():
    if not o.startswith("v_"):
        continue
    if "f64" in o and ("fma" in o or "mul" in o or "add" in o):
        grp["f64 arith"] += v
    elif o.startswith(("v_mov", "v_accvgpr")):
        grp["moves"] += v
    elif o.startswith("v_cndmask"):
        grp["cndmask"] += v
    elif o.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        grp["lane<->sgpr"] += v
    elif o.startswith("v_cmp"):
        grp["compares"] += v
    elif o.startswith(("v_xor", "v_and", "v_or", "v_lshl", "v_lshr", "v_ashr", "v_not", "v_bfe", "v_alignbit", "v_mad_u64", "v_mul_lo", "v_mul_hi", "v_add_u32", "v_sub_u32", "v_add_co", "v_addc", "v_lshl_add")):
        grp["int"] += v
    else:
        grp["other:" + o] += v
for g, v in grp.most_common(30):
    print(f"  {g:28s}{v}")
