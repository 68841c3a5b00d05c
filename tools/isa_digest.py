#!/usr/bin/env python3
"""Digest of the machine code of every kernel in the given objects (default: the no-diagnostics benchmark kernels and the team kernels).

    python3 tools/isa_digest.py [--write] [OBJ...]

The flagship kernel is sensitive, at the 1 % level, to source changes it never executes: in round 3 a descriptor copy in a prologue
cost 1.2 %, in round 4 spelling a compile-time constant of the non-team form as `G * T - 1` (G = 1) instead of `T - 1` reordered 3 500
of its 4 900 instructions and cost 0.8 % (profiles/round4/README.md).  The digests of the shipped benchmark kernels are therefore
RECORDED (profiles/round4/isa_digests.json, --write) and tests/test_build_invariants.py compares the build with the record: a kernel
whose code moved has to be re-measured before the record is rewritten."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("FS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
RECORD = os.path.join(ROOT, "profiles", "round4", "isa_digests.json")
DEFAULT = [os.path.join(ROOT, "flow-sim_amd", "csrc", "build", f) for f in ("fs_part_nodiag.o", "fs_part_team.o")]


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def digests(obj):
    with tempfile.TemporaryDirectory() as tmp:
        bundle, co = os.path.join(tmp, "f.bundle"), os.path.join(tmp, "f.co")
        run(f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={bundle}", obj, os.path.join(tmp, "copy.o"))
        run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}", f"--output={co}")
        out, name, body, pcrel = {}, None, [], 0
        for line in run(f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co).splitlines():
            m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
            if m:
                if name:
                    out[name] = body
                name, body = m.group(1), []
            elif name and line.strip():
                ins = re.sub(r"\s+", " ", line.split("//")[0].strip())
                # the displacement of a pc-relative address (s_getpc_b64, then s_add_u32 / s_addc_u32 with a literal: the kernel's constant tables)
                # depends on where the kernel sits in its object, i.e. on the SIZE OF ITS NEIGHBOURS: not part of the kernel's own code
                if pcrel and ins.startswith(("s_add_u32", "s_addc_u32")):
                    ins = re.sub(r", (0x[0-9a-f]+|-?\d+)$", ", REL", ins)
                    pcrel -= 1
                else:
                    pcrel = 2 if ins.startswith("s_getpc_b64") else 0
                body.append(ins)
        if name:
            out[name] = body
    names = run("c++filt", *out.keys()).splitlines()
    return {n.replace("fs::preissmann_step_kernel", "step").replace("(fs::KernelArgs<double>)", "").replace("(fs::KernelArgs<float>)", "").replace("void ", ""):
            dict(instructions=len(b), md5=hashlib.md5("\n".join(b).encode()).hexdigest()) for n, b in zip(names, out.values())}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--write"]
    got = {}
    for obj in (args or DEFAULT):
        got.update(digests(obj))
    if "--write" in sys.argv:
        json.dump(got, open(RECORD, "w"), indent=1, sort_keys=True)
    for k, v in sorted(got.items()):
        print(f"{v['md5'][:12]} {v['instructions']:6d}  {k}")
