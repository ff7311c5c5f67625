#!/usr/bin/env python3
"""End-to-end runner at scale (GPU box): writes N synthetic 8-kb reads as FASTQ under /tmp, runs
porechop_amd.runner over the file and prints the per-stage wall clock.
    python tools/runner_scale.py [n_reads]"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from porechop_amd import runner
from porechop_amd.synth import make_reads

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
L = 8000
reads = make_reads(n, L, seed=9, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
seq = reads.arena[: n * L].view(n, L).cpu().numpy()
name_w = 9
rec = np.empty((n, 1 + name_w + 1 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@")
digits = np.arange(n)[:, None] // (10 ** np.arange(name_w - 2, -1, -1))[None, :] % 10
rec[:, 1] = ord("r")
rec[:, 2:1 + name_w] = digits + ord("0")
c = 1 + name_w
rec[:, c] = 10
rec[:, c + 1:c + 1 + L] = seq
rec[:, c + 1 + L:c + 4 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, c + 4 + L:c + 4 + 2 * L] = ord("5")
rec[:, -1] = 10
path = "/tmp/runner_scale.fastq"
t = time.perf_counter()
rec.tofile(path)
print("wrote %s: %.2f GB in %.1f s" % (path, rec.size / 1e9, time.perf_counter() - t))
del rec, seq, reads
torch.cuda.empty_cache()
for rep in range(2):
    t = time.perf_counter()
    res = runner.run(path, output="/tmp/runner_scale_out.fastq")
    dt = time.perf_counter() - t
    print("run %d: %d reads in %.2f s = %.0f reads/s; sets %s; hits %d" % (rep, res.n_reads, dt, res.n_reads / dt, res.matching_sets, res.middle_hit_reads))
    print("   " + ", ".join("%s %.2f" % kv for kv in res.seconds.items()))
os.remove(path)
os.remove("/tmp/runner_scale_out.fastq")
