#!/usr/bin/env python3
"""Interleaved A/B of library builds on one GPU box (cdna_hip_programming.md rule 24: same device, interleaved rounds).
usage: tools/ab_bench.py ROUNDS name=path[:bench args] ...   (path '-' = the in-tree libpgas_hip.so)
Each round runs every variant once as its own process (python bench.py --cpu-steps 0 --steps 6); prints per-variant medians of
ms/sweep and of the two kernels' dispatch durations."""
import json
import os
import statistics
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
variants = []
for spec in sys.argv[2:]:
    name, rest = spec.split("=", 1)
    path, _, extra = rest.partition(":")
    variants.append((name, path, extra.split() if extra else []))
res = {v[0]: [] for v in variants}
for r in range(rounds):
    for name, path, extra in variants:
        env = dict(os.environ)
        for tok in [e for e in extra if "=" in e and not e.startswith("-")]:   # NAME=value tokens are environment settings
            k, v = tok.split("=", 1)
            env[k] = v
        extra = [e for e in extra if not ("=" in e and not e.startswith("-"))]
        if path != "-":
            env["PGAS_HIP_LIB"] = os.path.join(root, path)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-steps", "0", "--steps", "6"] + extra, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:], flush=True)
            continue
        j = json.loads(line[-1])
        rf = j.get("roofline", {})
        ks = {rf.get("kernel", "?"): rf.get("avg_launch_us"), rf.get("second_kernel", {}).get("kernel", "?"): rf.get("second_kernel", {}).get("avg_launch_us")}
        res[name].append((j["ms_per_step"], ks))
        print(f"round {r} {name:12s} {j['ms_per_step']:8.2f} ms/sweep  " + "  ".join(f"{k}: {v:.1f} us" for k, v in ks.items() if v), flush=True)
print("---- medians")
for name, rows in res.items():
    if rows:
        ms = statistics.median(x[0] for x in rows)
        keys = sorted({k for x in rows for k in x[1]})
        kk = "  ".join(f"{k}: {statistics.median(x[1][k] for x in rows if x[1].get(k)):.1f} us" for k in keys if any(x[1].get(k) for x in rows))
        print(f"{name:12s} {ms:8.2f} ms/sweep (min {min(x[0] for x in rows):.2f})  {kk}")
