"""Import-path compatibility: the reference's case scripts do `from src.hydromodel.channel import
Channel` etc. (cases/example/main.py:1-6).  These modules re-export the MI355X host mirror that
lives in flow-sim_amd/flowsim_amd/hydromodel."""
import os
import sys

_pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flow-sim_amd")
if _pkg not in sys.path:
    sys.path.insert(0, _pkg)
