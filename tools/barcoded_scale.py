#!/usr/bin/env python3
"""A native-barcoded run end to end (GPU box): N x 8 kb reads carrying one of 12 native barcodes
(Y adapter + flank + barcode on both ends, 10 % mutated), binned with -b: ~50 middle adapters,
half of them 63-68 bases long.     python tools/barcoded_scale.py [n_reads]"""
import os, shutil, sys, time
sys.path.insert(0, ".")
import numpy as np
from porechop_amd import runner
from porechop_amd.panel import load_panel, full_native_barcode

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
L = 8000
rng = np.random.default_rng(12)
panel = load_panel()
fulls = [full_native_barcode(panel, b) for b in range(1, 13)]
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
seq = acgt[rng.integers(0, 4, (n, L), dtype=np.uint8)]
bc = rng.integers(0, 12, n)
for i in range(12):
    rows = np.nonzero(bc == i)[0]
    s = np.frombuffer(fulls[i].start[1].encode(), dtype=np.uint8)
    e = np.frombuffer(fulls[i].end[1].encode(), dtype=np.uint8)
    seq[np.ix_(rows, np.arange(len(s)))] = s
    seq[np.ix_(rows, np.arange(L - len(e), L))] = e
# 8 % substitutions inside the adapter regions
mut = rng.random((n, 70)) < 0.08
sub = acgt[rng.integers(0, 4, (n, 70), dtype=np.uint8)]
seq[:, :70] = np.where(mut, sub, seq[:, :70])
mut = rng.random((n, 70)) < 0.08
seq[:, L - 70:] = np.where(mut, sub, seq[:, L - 70:])
name_w = 9
rec = np.empty((n, 1 + name_w + 1 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
rec[:, 2:1 + name_w] = np.arange(n)[:, None] // (10 ** np.arange(name_w - 2, -1, -1))[None, :] % 10 + ord("0")
c = 1 + name_w
rec[:, c] = 10; rec[:, c + 1:c + 1 + L] = seq
rec[:, c + 1 + L:c + 4 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, c + 4 + L:c + 4 + 2 * L] = ord("5"); rec[:, -1] = 10
path = "/tmp/barcoded.fastq"
rec.tofile(path)
del rec, seq
for rep in range(2):
    out = "/tmp/barcoded_bins"
    shutil.rmtree(out, ignore_errors=True)
    t = time.perf_counter()
    res = runner.run(path, barcode_dir=out)
    dt = time.perf_counter() - t
    calls = np.array(res.barcode_calls)
    want = np.array(["BC%02d" % (b + 1) for b in bc])
    print("run %d: %d reads in %.2f s = %.0f reads/s; %d sets (%s ...); correctly binned %.1f %%, unassigned %.1f %%" %
          (rep, n, dt, n / dt, len(res.matching_sets), ", ".join(res.matching_sets[:3]), 100 * (calls == want).mean(), 100 * (calls == "none").mean()))
    print("   " + ", ".join("%s %.2f" % kv for kv in res.seconds.items()))
os.remove(path); shutil.rmtree("/tmp/barcoded_bins", ignore_errors=True)
