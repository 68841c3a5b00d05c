"""The mirror package (src/hydromodel -> flowsim_amd.hydromodel) against the reference's public surface.

tests/golden/api_surface.json was written by oracle/gen_api_surface.py from the reference itself (imported in the build
container): every class, method and function a case script can reach, with parameter names, kinds and defaults.
SURVEY.md 8(b): the Channel / Boundary / Solver / plugin surface stays identical so that cases/* run unchanged.

What is compared: the names exist in the module a case script imports them from (`src.hydromodel.<module>`), and a call
written for the reference binds to the mirror's signature - same parameter names in the same order, same defaults, the
same kind.  The mirror may accept MORE (trailing optional parameters), never less.

The Newton-loop internals of PreissmannSolver (the per-entry Jacobian formulas, preissmann.py:200-897) are not Python
methods in this implementation - they are the HIP kernel - and are listed below as such, one by one, so that a method
that goes missing by accident still fails the test."""
import importlib
import inspect
import json
import os

import pytest

from conftest import GOLDEN

SURFACE = json.load(open(os.path.join(GOLDEN, "api_surface.json")))

# reference methods that are the body of the hot path: implemented by the kernel (flow-sim_amd/csrc), SURVEY 8(a) rows a3-a7
IN_KERNEL = {
    "PreissmannSolver": {
        "compute_residual_vector", "compute_jacobian", "compute_jacobian_data", "compute_indicies", "update_guesses",
        "upstream_residual", "downstream_residual", "continuity_residual", "momentum_residual",
        "dU_dh", "dU_dQ", "dD_dh", "dD_dQ",
        "dC_dh_i", "dC_dh_ip1", "dC_dQ_i", "dC_dQ_ip1", "dM_dh_i", "dM_dh_ip1", "dM_dQ_i", "dM_dQ_ip1",
        "time_diff", "spatial_diff", "cell_avg",
    },
}


def mirror_module(name):
    return importlib.import_module("src.hydromodel." + name)


def check_params(where, want, fn):
    got = list(inspect.signature(fn).parameters.values())
    assert len(got) >= len(want), f"{where}: takes {[p.name for p in got]}, the reference {[w[0] for w in want]}"
    for (name, kind, default), p in zip(want, got):
        assert p.name == name, f"{where}: parameter '{p.name}' where the reference has '{name}'"
        if kind in ("VAR_POSITIONAL", "VAR_KEYWORD"):
            assert p.kind.name == kind, f"{where}: {name} is {p.kind.name}, reference {kind}"
        if default is None:
            continue                              # required in the reference: required or defaulted here, both bind
        assert p.default is not inspect.Parameter.empty, f"{where}: '{name}' has a default in the reference ({default})"
        if not default.startswith("<"):
            assert repr(p.default) == default, f"{where}: default of '{name}' is {p.default!r}, reference {default}"
    for p in got[len(want):]:                     # extras must be optional
        assert p.default is not inspect.Parameter.empty or p.kind.name in ("VAR_POSITIONAL", "VAR_KEYWORD"), \
            f"{where}: extra required parameter '{p.name}'"


@pytest.mark.parametrize("module", sorted(SURFACE))
def test_module_functions_match_the_reference(module):
    mod = mirror_module(module)
    for name, params in SURFACE[module]["functions"].items():
        assert hasattr(mod, name), f"src.hydromodel.{module}.{name} is missing"
        if params is not None:
            check_params(f"{module}.{name}", params, getattr(mod, name))


CLASSES = [(m, c) for m in sorted(SURFACE) for c in sorted(SURFACE[m]["classes"])]


@pytest.mark.parametrize("module,cls", CLASSES)
def test_class_surface_matches_the_reference(module, cls):
    mod = mirror_module(module)
    assert hasattr(mod, cls), f"src.hydromodel.{module}.{cls} is missing"
    klass = getattr(mod, cls)
    ref = SURFACE[module]["classes"][cls]
    for base in ref["bases"]:
        assert base in [b.__name__ for b in klass.__mro__[1:]], f"{cls} does not derive from {base}"
    missing = []
    for mname, desc in ref["members"].items():
        if mname in IN_KERNEL.get(cls, ()):
            continue
        if not hasattr(klass, mname):
            missing.append(mname)
            continue
        member = inspect.getattr_static(klass, mname)
        if desc["kind"] == "property":
            continue                                 # readable attribute either way
        if desc["kind"] in ("staticmethod", "classmethod"):
            assert type(member).__name__ == desc["kind"], f"{cls}.{mname} is not a {desc['kind']}"
            member = member.__func__
        if desc["params"] is not None:
            check_params(f"{cls}.{mname}", desc["params"], member)
    assert not missing, f"{cls} lacks {missing}"


def test_in_kernel_list_names_only_reference_methods():
    for cls, names in IN_KERNEL.items():
        ref = next(SURFACE[m]["classes"][cls] for m in SURFACE if cls in SURFACE[m]["classes"])
        assert names <= set(ref["members"]), names - set(ref["members"])
