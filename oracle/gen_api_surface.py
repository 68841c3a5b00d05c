#!/usr/bin/env python3
"""API-surface fixture: imports the *reference* package (cve-mohd/flow-sim, /root/reference, read-only) in THIS
container and writes the names a case script can reach - classes, methods, functions, their parameter names, kinds and
defaults - to tests/golden/api_surface.json.

TEST INFRASTRUCTURE ONLY (like gen_golden.py): the reference cannot travel to the GPU box, the description of its public
surface can.  tests/test_api_surface.py holds the mirror (src/hydromodel -> flowsim_amd.hydromodel) against it:
SURVEY.md 8(b) asks for the Channel / Boundary / Solver / plugin surface to stay identical so that cases/* run unchanged.

    python oracle/gen_api_surface.py
"""
import importlib
import inspect
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
MODULES = ["boundary", "channel", "cross_section", "hydraulics", "hydrograph", "lumped_storage", "preissmann",
           "rating_curve", "solver", "utility"]          # lax.py: broken in the reference, out of scope


def default_repr(p):
    if p.default is inspect.Parameter.empty:
        return None
    d = p.default
    if isinstance(d, (int, float, str, bool, type(None))):
        return repr(d)
    return "<%s>" % type(d).__name__


def signature_of(fn):
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    return [[p.name, p.kind.name, default_repr(p)] for p in sig.parameters.values()]


def describe(mod):
    out = {"functions": {}, "classes": {}}
    for name, obj in vars(mod).items():
        if name.startswith("_") or getattr(obj, "__module__", None) != mod.__name__:
            continue
        if inspect.isfunction(obj):
            out["functions"][name] = signature_of(obj)
        elif inspect.isclass(obj):
            members = {}
            for mname, m in vars(obj).items():
                if mname.startswith("_") and mname != "__init__":
                    continue
                fn = m.__func__ if isinstance(m, (staticmethod, classmethod)) else m
                if inspect.isfunction(fn):
                    members[mname] = {"kind": type(m).__name__ if isinstance(m, (staticmethod, classmethod)) else "method",
                                      "params": signature_of(fn)}
                elif isinstance(m, property):
                    members[mname] = {"kind": "property", "params": None}
            out["classes"][name] = {"bases": [b.__name__ for b in obj.__bases__ if b is not object], "members": members}
    return out


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    surface = {}
    for m in MODULES:
        surface[m] = describe(importlib.import_module("src.hydromodel." + m))
    path = os.path.join(ROOT, "tests", "golden", "api_surface.json")
    with open(path, "w") as f:
        json.dump(surface, f, indent=1, sort_keys=True)
    n = sum(len(v["functions"]) + sum(len(c["members"]) for c in v["classes"].values()) for v in surface.values())
    print("wrote", path, "-", n, "callables in", len(surface), "modules")


if __name__ == "__main__":
    main()
