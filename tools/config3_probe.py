#!/usr/bin/env python3
"""BASELINE configs[2] shape (GPU box): phase B of N reads against the WHOLE panel's barcode sets
(96 forward barcodes x {start,end} + the ligation adapters), i.e. ~200 pairs per read, timed.
    python tools/config3_probe.py [n_reads]"""
import sys, time
sys.path.insert(0, ".")
import torch
from porechop_amd.panel import load_panel
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
panel = load_panel()
pl = Pipeline(panel, ScanParams())
reads = make_reads(n, 8000, seed=2, start_frac=0.9, end_frac=0.5)
matching = [i for i, s in enumerate(panel) if s.name == "SQK-NSK007" or ("Barcode" in s.name and "(forward)" in s.name)]
bc = {i for i in matching if "Barcode" in panel[i].name}
pl.aligner.set_timing(True)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    st, et, fulls = pl.phase_b(reads, matching, full_for=bc)
    pl.aligner.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    tm = pl.aligner.get_timing()
    pairs = n * 2 * len(matching)
    cells = n * 150 * sum(len(panel[i].start[1]) + len(panel[i].end[1]) for i in matching)
    print("rep %d: %d reads x %d sets: %.3f s wall, kernels %.1f ms, %.1f M pairs/s, %.2f TCUPS, %.0f reads/s; mem %.1f GB" %
          (rep, n, len(matching), dt, tm["trace"][0], pairs / dt / 1e6, cells / dt / 1e12, n / dt, torch.cuda.max_memory_allocated() / 1e9))
