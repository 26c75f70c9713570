"""Static instruction mix of selected kernels in csrc/pgas_api.s (make asm)."""
import re
import subprocess
import sys
from collections import Counter

path = sys.argv[1]
pats = sys.argv[2:]
lines = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for i, name in starts:
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if not any(p in dn for p in pats):
        continue
    body = []
    for l in lines[i + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        body.append(t)
    c = Counter(t.split()[0] for t in body)
    g = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))
    print(f"{dn[:48]:50s} total {len(body):6d}  f64 {sum(v for k, v in c.items() if 'f64' in k):5d}  mul32 {c['v_mul_hi_u32'] + c['v_mul_lo_u32']:4d}  "
          f"div_scale {c['v_div_scale_f64']:3d} rcp {c['v_rcp_f64']:3d} rsq {c['v_rsq_f64']:3d}  ds {g('ds_'):4d} (bperm {c['ds_bpermute_b32']})  "
          f"global {g('global_'):4d}  s_load {g('s_load'):4d}  salu {g('s_'):5d} waitcnt {c['s_waitcnt']}")
    print("    ", c.most_common(16))
