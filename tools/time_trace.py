#!/usr/bin/env python3
"""Kernel-time probe (GPU box): end windows (phase-B shape) in one traced pass vs the two-pass
scheme, and a whole-read scan, timed with the library's own HIP-event hooks.
   python tools/time_trace.py [n_reads]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import porechop_amd
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "GGTTGTTTCTGTTGGTGCTGATATTGCTGGCGTCTGCTT", "AAGCAGACGCCAGCAATATCAGCACCAACAGAAA"]
reads = make_reads(n, 8000, seed=5, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
al = porechop_amd.Aligner(ads)
al.set_timing(True)
wl = torch.full((n,), 150, dtype=torch.int32, device="cuda")
ref = {}
for mode, mname in ((porechop_amd.MODE_TRACE, "one traced pass"), (porechop_amd.MODE_TWO_PASS, "two-pass")):
    for name, off in (("start windows", reads.off), ("end windows", reads.off + 7850)):
        out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
        for rep in range(3):
            al.scan_device(reads.arena, off, wl, [0], [0, n], 150, out, mode, job_adapter_b=[1])
            al.sync()
            t = al.get_timing()
        tot = sum(v[0] for v in t.values())
        same = ""
        if name in ref:
            same = " identical=%s" % bool((ref[name] == out).all())
        ref.setdefault(name, out.clone())
        print("%-14s 150-col dual, %-16s %.2f ms  %s%s" % (name, mname, tot, {k: round(v[0], 2) for k, v in t.items() if v[1]}, same))
out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
for rep in range(2):
    al.scan_device(reads.arena, reads.off, reads.length, [2], [0, n], 8000, out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[3])
    al.sync()
    t = al.get_timing()
print("two-pass 8 kb dual:", {k: round(v[0], 2) for k, v in t.items() if v[1]})
