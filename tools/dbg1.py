import sys, random
sys.path.insert(0, '.')
import torch, numpy as np
from porechop_amd.pipeline import Pipeline, ScanParams, AdapterSet
from porechop_amd.synth import reads_from_strings
from tests.test_gpu_pipeline import make_reads, panel_sets
from oracle.oracle import Oracle
import porechop_amd
o = Oracle()
rng = random.Random(2024)
raw = make_reads(rng, 96)
pl = Pipeline(panel_sets(None), ScanParams())
dreads, norm = reads_from_strings(raw)
so, sl = pl._end_windows(dreads, None, "start")
eo, el = pl._end_windows(dreads, None, "end")
jobs = []; side=[]
for si, s in enumerate(pl.sets):
    if s.start is not None: jobs.append((pl.seq_index[s.start[1]], so, sl)); side.append(0)
    if s.end is not None: jobs.append((pl.seq_index[s.end[1]], eo, el)); side.append(1)
outs = pl._scan_jobs(dreads.arena, jobs, 1, 150)
try:
    pl.aligner.sync()
except Exception as e:
    print("SYNC ERR", e)
tot=0
for k, rec in enumerate(outs):
    rec = rec.cpu().numpy()
    ad = pl.seqs[jobs[k][0]]
    bad = 0
    for r in range(len(norm)):
        w = norm[r][:150] if side[k]==0 else norm[r][-150:]
        got = porechop_amd.format_result(rec[r]); want = o.adapter_alignment(w, ad)
        if got != want:
            bad += 1
            if bad < 2 and tot < 12: print(k, r, len(norm[r]), got, want)
    if bad: print("job", k, "adlen", len(ad), "adidx", jobs[k][0], "bad", bad); tot+=1
print("bad jobs", tot, "of", len(jobs))
