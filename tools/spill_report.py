#!/usr/bin/env python3
"""Registers, scratch and spill counts of every kernel of the built library (llvm-readelf notes of the gfx950 code objects bundled in the objects).

    python3 tools/spill_report.py [OBJ...]          (default: flow-sim_amd/csrc/build/*.o)

The kernels that spill BOTH scalar registers (into VGPR lanes) and vector registers (to scratch) sit on the register allocator's edge: a change to shared
code moves their spill code around, and round 4's unexplained fault was in one of them (profiles/round4/irr_strip_k15_reverted.txt)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("FS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        bundle, co = os.path.join(tmp, "f.bundle"), os.path.join(tmp, "f.co")
        try:
            run(f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={bundle}", obj, os.path.join(tmp, "copy.o"))
            run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}", f"--output={co}")
        except subprocess.CalledProcessError:
            return []
        out, cur = [], {}
        for line in run(f"{LLVM}/llvm-readelf", "--notes", co).splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.+)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur.get("name"):
                out.append(cur); cur = {}
            if k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                cur[k] = v
        if cur.get("name"):
            out.append(cur)
        return out


if __name__ == "__main__":
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "flow-sim_amd", "csrc", "build", "fs_part_*.o")))
    rows = []
    for o in objs:
        for k in kernels(o):
            name = run("c++filt", k["name"]).strip()
            name = re.sub(r"^void fs::preissmann_(step|long)_kernel", r"\1", name).split("(fs::KernelArgs")[0]
            rows.append((int(k.get("sgpr_spill_count", 0)), int(k.get("vgpr_spill_count", 0)), int(k.get("private_segment_fixed_size", 0)),
                         int(k.get("vgpr_count", 0)), int(k.get("group_segment_fixed_size", 0)), os.path.basename(o), name))
    print(f"{'SGPR spills':>11} {'VGPR spills':>11} {'scratch B':>9} {'registers':>9} {'LDS B':>7}  kernel")
    for r in sorted(rows, key=lambda r: (-(r[0] > 0 and r[1] > 0), -r[1], -r[0])):
        print(f"{r[0]:11d} {r[1]:11d} {r[2]:9d} {r[3]:9d} {r[4]:7d}  {r[6]}   [{r[5]}]")
    both = sum(1 for r in rows if r[0] > 0 and r[1] > 0)
    print(f"# {len(rows)} kernels, {both} of them spill both scalar and vector registers")
