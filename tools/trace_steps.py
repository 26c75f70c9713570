"""Per-step kernel durations from a rocprofv3 --kernel-trace CSV (development aid)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
for name in ("k_resample", "void k_upper", "void k_propagate", "void k_fused"):
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"].startswith(name)])
    if not len(d):
        continue
    print(f"{name}: n={len(d)} mean={d.mean():.2f} us  median={np.median(d):.2f}  min={d.min():.2f} max={d.max():.2f}")
    if len(d) > 100:
        n = len(d)
        per = n // (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
        seg = d[-per:]
        edges = [0, 5, 20, 50, 100, 200, 400, 800, 1200, 1600, per]
        for a, b in zip(edges, edges[1:]):
            if a < per:
                print(f"   launches {a:5d}-{min(b,per):5d}: mean {seg[a:b].mean():7.2f}  max {seg[a:b].max():7.2f}")
