"""Average PMC counters per kernel from a rocprofv3 --pmc run directory (development aid).
usage: pmc_kernels.py DIR [raw]   -- default: per wave; raw: per dispatch."""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
raw = len(sys.argv) > 2 and sys.argv[2] == "raw"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    grid[k] = int(r["Grid_Size"]) // 64
for k, d in acc.items():
    if k.startswith(("k_", "void k_")):
        if raw:
            print(k, "dispatches", len(next(iter(d.values()))), {c: round(sum(v) / len(v), 1) for c, v in d.items()})
        else:
            print(k, "waves", grid[k], {c: round(sum(v) / len(v) / grid[k], 1) for c, v in d.items()})
