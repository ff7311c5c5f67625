#!/usr/bin/env python3
"""The PCIe-inclusive leg of bench.py alone (GPU box): reads start in pinned host memory, every step uploads them again in
batches while the previous batch is scanned.   python tools/h2d_leg.py   (PC_BENCH_H2D_BATCHES=n changes the batch count)"""
import sys, json, argparse
sys.path.insert(0, ".")
import torch, bench
args = argparse.Namespace(gpus=1, steps=5, warmup=2, reads=1_000_000, reads1=100_000, reads2=1_000_000, read_len=8000, chimera=0.01, cpu_seconds=0.0, no_extra=False)
print(json.dumps(bench.leg_host_buffers(torch.device("cuda", 0), args)))
