"""Audit of the exactness gates of the 16-bit kernels (tests/host/test_bounds.cpp, host-compiled against
porechop_amd/csrc/pc_bounds.h): over a grid of 1.5 M (scoring scheme, row class) combinations, whenever
f16_plan / spec_plan admit the packed-fp16 kernels, every quantity those kernels form -- bounded independently
from the recurrence -- must be an integer fp16 holds exactly.  No GPU involved."""
import os
import subprocess
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp16_gates_admit_only_exact_schemes():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "test_bounds")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I", os.path.join(REPO, "porechop_amd", "csrc"),
                               os.path.join(REPO, "tests", "host", "test_bounds.cpp"), "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "bad=0" in out.stdout, out.stdout[-3000:]
        fields = dict(kv.split("=") for kv in out.stdout.strip().splitlines()[-1].split())
        assert int(fields["checked"]) > 1_000_000 and int(fields["admitted_trace16"]) > 100_000 and int(fields["admitted_spec"]) > 100_000
