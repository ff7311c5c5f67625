#!/usr/bin/env python3
"""Compile the run-time specialised kernel source OFFLINE (hipcc -S) for one adapter pair, to look
at its ISA / register use without a GPU.   python tools/dump_spec.py [--int16] [AD_LO AD_HI] > x.s"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "..", "porechop_amd", "csrc", "pc_jit_source.h")).read()
src = src[src.index('R"PCJIT(') + 8: src.index(')PCJIT"')]
args = [a for a in sys.argv[1:] if not a.startswith("--")]
f16 = "--int16" not in sys.argv
lo, hi = (args + ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"])[:2]
match, mismatch, go, ge = 3, -6, -5, -2
R = max(len(lo), len(hi))
code = {"A": 0, "C": 1, "G": 2, "T": 3}
L = [5] * (R - len(lo)) + [code.get(c, 4) for c in lo]
H = [5] * (R - len(hi)) + [code.get(c, 4) for c in hi]
combos, rows = [], []
for l, h in zip(L, H):
    c = l * 6 + h
    if c not in combos: combos.append(c)
    rows.append(combos.index(c))
K = (len(combos) + 3) // 4 * 4
eps = -ge
low = min(2 * go + (R - 1) * ge, go + (R - 1) * ge + mismatch, go)
high = match * R
lim = 2040 if f16 else 32000
kren = min((2 * lim - (high - low) - (R + 6) * eps) // eps // 4 * 4, 1 << 20)
defs = ["-DPC_R=%d" % R, "-DPC_K=%d" % K, "-DPC_COMBO_INIT=" + ",".join(map(str, rows)), "-DPC_F16=%d" % f16,
        "-DPC_EPS=%d" % eps, "-DPC_OE=(%d)" % (go + eps), "-DPC_CEN=(%d)" % (low + lim), "-DPC_KREN=%d" % kren, "-DPC_WAVES=%s" % __import__("os").environ.get("PC_JIT_WAVES", "2"), "-DPC_CHECK_RANGE=%d" % ("--check" in sys.argv), "-DPC_DUAL=%d" % (lo != hi)]
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "k.hip")
    open(p, "w").write("#include <hip/hip_runtime.h>\n" + src)
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                          "-o", "-", p] + defs, capture_output=True, text=True)
    sys.stderr.write(out.stderr)
    sys.stdout.write(out.stdout)
    sys.stderr.write("R=%d K=%d kren=%d %s\n" % (R, K, kren, " ".join(defs)))
