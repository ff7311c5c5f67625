#!/usr/bin/env python3
"""Host-side cProfile of one extra leg of bench.py (GPU box):  python tools/profile_leg.py ragged|configs1|configs2"""
import argparse, cProfile, pstats, sys, io
sys.path.insert(0, ".")
import torch
import bench
leg = sys.argv[1] if len(sys.argv) > 1 else "ragged"
args = argparse.Namespace(gpus=1, steps=5, warmup=2, reads=1_000_000, reads1=100_000, reads2=1_000_000, read_len=8000, chimera=0.01,
                          cpu_seconds=0.0, no_extra=False)
dev = torch.device("cuda", 0)
fn = {"ragged": lambda: bench.leg_ragged(dev, args, 1, 73e9), "configs1": lambda: bench.leg_configs1(dev, args, 1),
      "configs2": lambda: bench.leg_configs2(dev, args, 1)}[leg]
def headline():
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads
    pl = Pipeline(bench.load_panel_sets(), ScanParams(), device=dev)
    reads = make_reads(args.reads, args.read_len, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev)
    for _ in range(8):
        bench.one_step(pl, reads, 10000, 1)
    for _ in range(3):
        bench.one_step(pl, reads, 10000, 1, proofs=True)
    pl.aligner.sync(); pl.close()
    return {"reads_per_s": 0}
for pre in sys.argv[2:]:          # legs to run first, unprofiled (the state they leave behind is part of the question)
    print("pre-leg", pre, {k: v for k, v in {"configs1": lambda: bench.leg_configs1(dev, args, 1), "configs2": lambda: bench.leg_configs2(dev, args, 1), "headline": lambda: headline()}[pre]().items() if k in ("reads_per_s",)})
    torch.cuda.empty_cache()
pr = cProfile.Profile()
pr.enable()
out = fn()
pr.disable()
print({k: v for k, v in out.items() if k in ("reads_per_s", "ms_per_step", "bp_per_s_vs_uniform_lengths")})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
