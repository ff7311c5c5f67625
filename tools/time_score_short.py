"""Score-only pass on 150-column end windows: specialised vs generic kernels (diagnostic, GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import porechop_amd
from porechop_amd.synth import make_reads
n = 1_000_000
WIN = int(os.environ.get("WIN", "150"))
from porechop_amd import _lib
_lib.load_library().pc_jit_async(0)
ads = ["AATGTACTTCGTTCAGTTACGTATTGCT", "GCAATACGTAACTGAACGAAGT", "AAGAAAGTTGTCGGTGTCTTTGTG", "TCGATTCCGTTTGTAGTCGTCTGT"]
reads = make_reads(n, 8000, seed=5, start_frac=0.9, end_frac=0.5)
al = porechop_amd.Aligner(ads)
al.set_timing(True)
base_off = reads.off
arena = reads.arena
if os.environ.get("DENSE"):
    stride = int(os.environ["DENSE"])
    arena = reads.arena[: n * stride + 4096].contiguous()
    base_off = torch.arange(n, dtype=torch.int64, device="cuda") * stride
wl = torch.full((n,), WIN, dtype=torch.int32, device="cuda")
for name, ja, jb in (("single 24-mer", [2], None), ("dual 24+24", [2], [3]), ("two singles in one call", [2, 3], None), ("single 28-mer", [0], None)):
    k = len(ja)
    out = torch.zeros((n * k * (2 if jb else 1), 8), dtype=torch.int32, device="cuda")
    starts = np.arange(k + 1, dtype=np.int64) * n
    off = torch.cat([base_off] * k); ln = torch.cat([wl] * k)
    for rep in range(3):
        al.scan_device(arena, off, ln, ja, starts, WIN, out, porechop_amd.MODE_SCORE, job_adapter_b=jb)
        al.sync()
        t = al.get_timing()
    cells = n * WIN * sum(len(ads[a]) for a in ja) + (n * WIN * sum(len(ads[b]) for b in jb) if jb else 0)
    tot = sum(v[0] for v in t.values())
    print("%-24s %.2f ms  %s  %.2f TCUPS" % (name, tot, {a: (round(b[0], 2), b[1]) for a, b in t.items() if b[1]}, cells / tot / 1e9))
