#!/usr/bin/env python3
"""Whole-read scan against LONG adapters (full native-barcode sequences, 63-67 bases; 111 bases):
the score pass these take (generic LDS-state kernel vs specialised).   python tools/time_long.py [n]"""
import sys
sys.path.insert(0, ".")
import torch
import porechop_amd
from porechop_amd.panel import load_panel, full_native_barcode, full_rapid_barcode_new
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
panel = load_panel()
nb1 = full_native_barcode(panel, 1)
rb1 = full_rapid_barcode_new(panel, 1)
ads = [nb1.start[1], nb1.end[1], rb1.start[1], rb1.start[1][:100]]
print([len(a) for a in ads])
reads = make_reads(n, 8000, seed=5, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
al = porechop_amd.Aligner(ads)
al.set_timing(True)
for (a, b) in ((0, 1), (2, 3)):
    out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
    for rep in range(2):
        al.scan_device(reads.arena, reads.off, reads.length, [a], [0, n], 8000, out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[b])
        al.sync()
        t = al.get_timing()
    cells = n * 8000 * (len(ads[a]) + len(ads[b]))
    ms = t["score"][0] + t["score_spec"][0]
    print("adapters %d+%d rows: score %.1f ms (%s) = %.2f TCUPS; trace %.1f ms" %
          (len(ads[a]), len(ads[b]), ms, "specialised" if t["score_spec"][1] else "generic", cells / ms / 1e9, t["trace"][0]))
