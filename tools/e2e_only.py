#!/usr/bin/env python3
"""Just the end-to-end (file -> file) leg of bench.py (GPU box):  python tools/e2e_only.py [reads]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
args = argparse.Namespace(reads_e2e=n, read_len=8000, chimera=0.01)
out = bench.leg_end_to_end(torch.device("cuda", 0), args)
print(json.dumps(out, indent=1))
