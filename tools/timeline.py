"""Chain timeline from a rocprofv3 --kernel-trace CSV (development aid): per step, the durations of k_step / k_groups / k_propagate and
the gaps between dependent launches on the chain stream.
usage: timeline.py DIR"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = {"k_step": [], "k_groups": [], "k_propagate": []}
for r in rows:
    n = r["Kernel_Name"]
    for k in ev:
        if k + "<" in n or n.startswith(k + "(") or (k == "k_groups" and n.startswith("k_groups_abs")):
            ev[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k in ev:
    ev[k].sort()
    a = np.array(ev[k])
    print(f"{k}: {len(a)} dispatches, duration median {np.median(a[:,1]-a[:,0])/1e3:.2f} us, mean {np.mean(a[:,1]-a[:,0])/1e3:.2f} us")
s, g, p = (np.array(ev[k]) for k in ("k_step", "k_groups", "k_propagate"))
n = min(len(s), len(g))
# sweeps are separated by long gaps; keep pairs inside a sweep
period = np.diff(s[:, 0]) / 1e3
inside = period < 200
print(f"k_step start-to-start period: median {np.median(period[inside]):.2f} us, mean {np.mean(period[inside]):.2f} us")
# match each k_groups to the preceding k_step
gi = np.searchsorted(s[:, 1], g[:, 0], side="right") - 1
ok = gi >= 0
gap1 = (g[ok, 0] - s[gi[ok], 1]) / 1e3
print(f"gap k_step end -> k_groups start: median {np.median(gap1):.2f} us, p90 {np.percentile(gap1, 90):.2f}")
si = np.searchsorted(g[:, 1], s[:, 0], side="right") - 1
ok = si >= 0
gap2 = (s[ok, 0] - g[si[ok], 1]) / 1e3
gap2 = gap2[gap2 < 200]
print(f"gap k_groups end -> next k_step start: median {np.median(gap2):.2f} us, p90 {np.percentile(gap2, 90):.2f}")
pp = np.diff(p[:, 0]) / 1e3
print(f"k_propagate start-to-start: median {np.median(pp[pp < 200]):.2f} us; gap end->next start median {np.median((p[1:,0]-p[:-1,1])[pp < 200])/1e3:.2f} us")
# overlap: fraction of k_step time during which a k_propagate is running
def covered(a, b):
    tot = 0
    j = 0
    for s0, s1 in a:
        while j < len(b) and b[j, 1] <= s0:
            j += 1
        k = j
        while k < len(b) and b[k, 0] < s1:
            tot += max(0, min(s1, b[k, 1]) - max(s0, b[k, 0]))
            k += 1
    return tot
print(f"fraction of k_step time with a k_propagate in flight: {covered(s, p) / np.sum(s[:,1]-s[:,0]):.2f}; of k_propagate time with a k_step in flight: {covered(p, s) / np.sum(p[:,1]-p[:,0]):.2f}")
# lag of the chain behind pipeline A: which propagate step is running when k_step t starts (indices within the last sweep)
