#!/usr/bin/env python3
"""GPU box: bench.py's end_to_end_gz leg alone.   python tools/e2e_gz_only.py [reads]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
class A: pass
a = A()
a.reads_e2e = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
a.read_len, a.chimera, a.cli_reads, a.cpu_seconds = 8000, 0.01, 0, 0.0
out = bench.leg_end_to_end_gz(torch.device("cuda", 0), a, bench.host_cores())
print(json.dumps({k: v for k, v in out.items() if k != "by_input_layout"}, indent=1))
print(json.dumps({k: {"reads_per_s": v["reads_per_s"], "stage": v["runs"][-1]["stage_seconds"], "md5": v["gunzipped_output_md5_equals_plain_route"]} for k, v in out["by_input_layout"].items()}, indent=1))
