#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy per kernel of one .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-munsafe-fp-atomics", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:], "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    print(f"{name:60s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('SGPRs','?'):>4} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} "
          f"lds={r.get('LDS Size [bytes/block]','?')}")
