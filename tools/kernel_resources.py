"""Condense hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin) into one line per kernel."""
import re
import sys

cur = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
    else:
        cur[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            print(f"{cur['name'][:70]:70s} VGPR={cur.get('VGPRs','?'):>4} SGPR={cur.get('SGPRs','?'):>4} scratch={cur.get('ScratchSize','?'):>4} "
                  f"occ={cur.get('Occupancy','?'):>2} LDS={cur.get('LDS','?'):>7}")
