"""Registers / spills / LDS of every kernel of a translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2>&1 | python tools/resource_usage.py [filter]"""
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur, rows = None, []
for line in sys.stdin:
    if "error" in line:
        print(line.rstrip())
    m = re.search(r"remark: [^ ]+ +(Function Name|Name): (\S+)", line) or re.search(r":\s+(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}; rows.append(cur); continue
    m = re.search(r":\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
import subprocess
for r in rows:
    try:
        nm = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        nm = r["name"]
    nm = nm.replace("fs::preissmann_step_kernel", "step").replace("(fs::KernelArgs<double>)", "").replace("(fs::KernelArgs<float>)", "").replace("void ", "")
    if flt and flt not in nm:
        continue
    print(f"{nm:60s} vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', -1):3d} scratch {r.get('ScratchSize', -1):4d} spill {r.get('VGPRs Spill', -1):3d} "
          f"occ {r.get('Occupancy', -1)} lds {r.get('LDS Size', -1)}")
