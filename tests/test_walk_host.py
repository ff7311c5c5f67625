"""Compiles porechop_amd/csrc/pc_walk.h + pc_bounds.h for the HOST and checks the traceback/digest
code the kernels run per lane, and the two-pass window bound, against the oracle on 20 000 seeded
cases (tests/host/test_walk.cpp).  No GPU involved."""
import os
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_walk_and_window_bound_against_oracle():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_walk")
        obj = os.path.join(tmp, "pc_oracle.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-c", os.path.join(REPO, "oracle", "pc_oracle.c"), "-o", obj])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(REPO, "porechop_amd", "csrc"),
                               "-I", os.path.join(REPO, "oracle"), os.path.join(REPO, "tests", "host", "test_walk.cpp"),
                               obj, "-o", exe])
        out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:]
        assert "bad=0" in out.stdout
        windows = int(out.stdout.split("windows_checked=")[1].split()[0])
        assert windows > 1000      # the bounded-window path really was exercised
