#!/usr/bin/env python3
"""GPU box: bench.py's from_host_memory leg alone (the step with the reads starting in pinned host memory).
    python tools/r6_h2d.py        (PC_FORK_STREAMS, GPU_MAX_HW_QUEUES etc. from the environment)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(reads=1_000_000, read_len=8000, chimera=0.01, steps=3, warmup=1)
v = bench.leg_host_buffers(torch.device("cuda", 0), args)
pf = v.get("exact_prefilter") or {}
print(json.dumps({"fork": os.environ.get("PC_FORK_STREAMS", ""), "hwq": os.environ.get("GPU_MAX_HW_QUEUES", ""),
                  "ms_per_step": round(v["ms_per_step"], 1), "prefilter_ms": round(pf.get("ms_per_step", 0.0), 1),
                  "bytes1_ms": round((v.get("bytes_per_base_1") or {}).get("ms_per_step", 0.0), 1)}))
