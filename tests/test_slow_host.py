"""Compiles porechop_amd/csrc/pc_slow.h (+ pc_walk.h) for the HOST and checks the plain-int32 alignment the HIP library
runs for scoring schemes / adapter lengths its packed 16-bit kernels refuse against the oracle on 30 000 seeded cases
over UNRESTRICTED integer schemes (tests/host/test_slow.cpp).  The same align_pair() the device runs per lane; no GPU."""
import os
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_int32_alignment_against_oracle_on_unrestricted_schemes():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_slow")
        obj = os.path.join(tmp, "pc_oracle.o")
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-c", os.path.join(REPO, "oracle", "pc_oracle.c"), "-o", obj])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(REPO, "porechop_amd", "csrc"),
                               "-I", os.path.join(REPO, "oracle"), os.path.join(REPO, "tests", "host", "test_slow.cpp"),
                               obj, "-o", exe])
        out = subprocess.run([exe, "30000"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:]
        assert "bad=0" in out.stdout
        assert int(out.stdout.split("nonnegative_gap=")[1].split()[0]) > 5000
        assert int(out.stdout.split("linear=")[1].split()[0]) > 2000
