#!/usr/bin/env python3
"""GPU box: what the traced end-window kernel spends where -- the phase-B shape (1 M start windows x one 28-mer, 1 M end
windows x one 22-mer, single-adapter jobs) under PC_DEBUG_TRACE = 0 (all), 1 (no traceback), 2 (slab stores to one column),
3 (both).  Run once per setting (the knob is read once):  for d in 0 1 2 3; do PC_DEBUG_TRACE=$d python tools/time_trace_parts.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import porechop_amd
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
reads = make_reads(n, 8000, seed=5, start_frac=0.9, end_frac=0.5, chimera_frac=0.0)
al = porechop_amd.Aligner(ads)
al.set_timing(True)
wl = torch.full((n,), 150, dtype=torch.int32, device="cuda")
for name, off, ad in (("start x 28-mer", reads.off, 0), ("end x 22-mer", reads.off + 7850, 1)):
    out = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
    best = 1e9
    for rep in range(int(os.environ.get('PC_LOOP', '4'))):
        al.scan_device(reads.arena, off.contiguous(), wl, [ad], [0, n], 150, out, porechop_amd.MODE_TRACE)
        try:
            al.sync()
        except Exception:
            pass
        t = al.get_timing()
        best = min(best, t["trace"][0])
    cells = n * 150 * len(ads[ad])
    print("PC_DEBUG_TRACE=%s %-15s %.3f ms  %.2f TCUPS (ceiling 5.9)" % (os.environ.get("PC_DEBUG_TRACE", "0"), name, best, cells / best / 1e9))
