"""The rating row's x^b (fs_device.hpp: log_pos, exp_short, pow_pos - the round-4 replacement of the two libm calls) as a CPU model of
the same operations (tools/micro/pow_model.c: fma() for every device fma, the reciprocal seed's error put in by hand) against powl()
on a random grid of 4e6 points, x in 1e-4 .. 1e4, b in 0.2 .. 5: no worse than libm's exp(b log x) on the same grid, log(x) to
4e-16 also next to 1.  (The device function itself runs in every parity test with a power rating curve: c5_512, the instantiation
and random sweeps.)"""
import os
import re
import subprocess

from conftest import ROOT


def test_the_written_out_power_is_as_accurate_as_the_libm_pair(tmp_path):
    exe = tmp_path / "pow_model"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tools", "micro", "pow_model.c"), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"worst rel err new ([0-9.e+-]+) .*libm exp\(b log x\) ([0-9.e+-]+)\s+log rel err ([0-9.e+-]+)", out)
    n = re.search(r"log near 1: worst rel ([0-9.e+-]+)", out)
    assert m and n, out
    new, old, log_err, near1 = float(m.group(1)), float(m.group(2)), float(m.group(3)), float(n.group(1))
    assert new <= 6e-15 and new <= 1.25 * old, out          # the rounding of b log x dominates both
    assert log_err <= 4e-16 and near1 <= 4e-16, out
