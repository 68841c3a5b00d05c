#!/usr/bin/env python3
"""Build invariant of libflowsim_hip.so: NO device function is called out of line.

    python3 tools/check_isa.py flow-sim_amd/csrc/build/*.o        (the Makefile runs it between compiling and linking)

Why (profiles/round3/polyline_calls.txt): in round 3 the inliner left one grown `__device__` function of the
polyline kernels out of line at one call site; the caller then trusted an interprocedural register summary of the callee that
did not hold - wrong numbers from the second level on in one instantiation, a GPU fault at first launch in another, depending on
unrelated code around the call.  `always_inline` on the kernel lambdas fixed that instance; this check keeps the next grown
function from repeating it.  For every gfx950 code object bundled in the given objects it fails when

  * a function symbol is not a kernel (no `<name>.kd` descriptor next to it): a device function was emitted out of line;
  * the disassembly holds `s_swappc_b64` or `s_call_b64` (a call; `s_setpc_b64` alone is only the long form of a branch);
  * a kernel's metadata says `.uses_dynamic_stack: true` (stack use the compiler could not bound: recursion or a call).

Prints one line per object and exits 1 on the first kind of violation found anywhere.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("FS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def check_object(obj, tmp):
    bundle, co = os.path.join(tmp, "f.bundle"), os.path.join(tmp, "f.co")
    run(f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={bundle}", obj, os.path.join(tmp, "copy.o"))      # (no output name: objcopy rewrites its input)
    run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={bundle}", f"--output={co}")
    problems = []
    funcs, descriptors = [], set()
    for line in run(f"{LLVM}/llvm-readelf", "-s", "--wide", co).splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC" and f[6] != "UND":
            funcs.append(f[7])
        elif len(f) >= 8 and f[3] == "OBJECT" and f[7].endswith(".kd"):
            descriptors.add(f[7][:-3])
    for name in funcs:
        if name not in descriptors:
            problems.append(f"device function emitted out of line: {name}")
    current, calls = None, {}
    for line in run(f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            current = m.group(1)
        elif "s_swappc_b64" in line or "s_call_b64" in line:
            calls[current] = calls.get(current, 0) + 1
    for name, n in calls.items():
        problems.append(f"{n} call instruction(s) (s_swappc_b64 / s_call_b64) in {name}")
    kernel = None
    for line in run(f"{LLVM}/llvm-readelf", "--notes", co).splitlines():
        line = line.strip()
        if line.startswith(".name:"):
            kernel = line.split(":", 1)[1].strip()
        elif line.startswith(".uses_dynamic_stack:") and line.endswith("true"):
            problems.append(f"dynamic stack in {kernel}")
    return len(descriptors), problems


def main(objs):
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            n, problems = check_object(obj, tmp)
            print(f"check_isa: {os.path.basename(obj)}: {n} kernels, " + ("no call out of line" if not problems else f"{len(problems)} VIOLATION(S)"))
            for p in problems:
                print("   ", p)
            bad += len(problems)
    if bad:
        print("check_isa: FAILED - a device function is called out of line (mark it / its caller's lambdas always_inline); the library is "
              "not linked.  See DESIGN.md section 4.4.")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
