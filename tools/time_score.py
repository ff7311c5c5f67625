#!/usr/bin/env python3
"""Score-pass probe (GPU box): pc_spec_score time per tile for batch sizes that do / do not divide
evenly over the persistent grid (tail effect).   python tools/time_score.py"""
import sys
sys.path.insert(0, ".")
import torch
import porechop_amd
from porechop_amd.synth import make_reads
ads = sys.argv[1:3] if len(sys.argv) > 2 else ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT"]
al = porechop_amd.Aligner(ads)
al.set_timing(True)
for n in ((1_000_000,) if len(sys.argv) > 2 else (2048 * 7 * 64, 1_000_000, 2048 * 8 * 64, 2048 * 4 * 64 + 64 * 100)):
    reads = make_reads(n, 8000, seed=5, start_frac=0.9, end_frac=0.5, chimera_frac=0.01)
    out = torch.zeros((2 * n, 8), dtype=torch.int32, device="cuda")
    for rep in range(3):
        al.scan_device(reads.arena, reads.off, reads.length, [0], [0, n], 8000, out, porechop_amd.MODE_TWO_PASS, job_adapter_b=[1])
        al.sync()
        t = al.get_timing()
    ms = t["score_spec"][0]
    tiles = (n + 63) // 64
    print("n=%8d tiles=%6d (%.3f per resident wave)  score %.2f ms  %.4f us/tile  trace %.2f ms" % (n, tiles, tiles / 2048, ms, ms * 1e3 / tiles, t["trace"][0]))
    del reads, out
    torch.cuda.empty_cache()
