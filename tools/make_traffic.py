"""Build profiles/traffic_<round>.json from the per-kernel PMC summaries tools/evidence.sh and tools/pmc_sq.sh wrote (development aid).
usage: make_traffic.py gpurun_out/r02 profiles/traffic_r02.json"""
import ast, json, re, sys, time, os
src, dst = sys.argv[1], sys.argv[2]


def read(name):
    out = {}
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return out
    for line in open(p):
        m = re.match(r"^(?:void )?(k_\w+).*? dispatches (\d+) (\{.*\})\s*$", line.strip())
        if m and int(m.group(2)) > out.get(m.group(1), {}).get("dispatches", 0):   # several instantiations: keep the one the time loop launches
            out[m.group(1)] = dict(dispatches=int(m.group(2)), **ast.literal_eval(m.group(3)))
    return out


fetch, write = read("pmc_FETCH_SIZE.txt"), read("pmc_WRITE_SIZE.txt")
insts, cycles, grbm, mix = read("pmc_sq_insts.txt"), read("pmc_sq_cycles.txt"), read("pmc_sq_grbm.txt"), read("pmc_sq_lds.txt")


def hbm(k):   # KiB -> bytes; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section)
    if k not in fetch or k not in write:
        return None
    return int(round(1024 * (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"])))


N = 1 << 20
out = {
    "source": "rocprofv3 --pmc <counters> --kernel-trace, separate passes, python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-profile --T 200 "
              "(tools/evidence.sh, tools/pmc_sq.sh); per-dispatch averages; FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled for gfx950",
    "collected": time.strftime("%Y-%m-%d"),
    "raw": {"FETCH_SIZE_KiB": {k: v["FETCH_SIZE"] for k, v in fetch.items()}, "WRITE_SIZE_KiB": {k: v["WRITE_SIZE"] for k, v in write.items()}},
    "k_step_hbm_bytes_per_launch": hbm("k_step"),
    "k_groups_hbm_bytes_per_launch": hbm("k_groups"),
    "k_propagate_hbm_bytes_per_step": hbm("k_propagate"),
    "algorithmic_bytes_per_step": 52 * N,
}
valu = {}
for k in ("k_propagate", "k_step", "k_groups"):
    if k not in insts:
        continue
    d = dict(insts[k])
    d.pop("dispatches", None)
    e = {"per_dispatch": d}
    if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
        e["valu_instr_per_wave"] = d["SQ_INSTS_VALU"] / max(d["SQ_WAVES"], 1)
        e["valu_instr_per_particle_step"] = 64 * d["SQ_INSTS_VALU"] / N if k != "k_groups" else None
        # an fp64 VALU instruction occupies its SIMD for 16 cycles per wave64 (4 passes of 16 lanes at 1/4 ... see DESIGN section 5);
        # the issue-bound time below uses the measured busy cycles instead of that model
    if k in cycles:
        c = dict(cycles[k]); c.pop("dispatches", None)
        e["cycles_per_dispatch"] = c
        if c.get("SQ_WAVE_CYCLES"):
            e["active_valu_over_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
            e["wait_any_over_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if k in grbm:
        g = dict(grbm[k]); g.pop("dispatches", None)
        e["grbm"] = g
    if k in mix:
        m = dict(mix[k]); m.pop("dispatches", None)
        e["mix_per_dispatch"] = m
    valu[k] = e
out["valu"] = valu
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
