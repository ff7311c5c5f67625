#!/usr/bin/env python3
"""Stage timings of the BASELINE configs[2] step (GPU box): phase A, kit choice, scan of phase B, reduce.
    python tools/demux_times.py [n_reads]"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from bench import load_panel_sets, load_panel_json
from porechop_amd import panel as rules
from porechop_amd.pipeline import Pipeline, ScanParams, MODE_TRACE
from porechop_amd.runner import Options, barcode_bins
from porechop_amd.synth import make_reads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p, opts = ScanParams(), Options()
pl = Pipeline(load_panel_sets(), p)
fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
reads = make_reads(n, 8000, seed=2, barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
pl.aligner.set_timing(True)
def T():
    pl.aligner.sync(); torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T()
    bs, be = pl.phase_a(reads, torch.arange(p.check_reads, device="cuda"))
    matching = pl.matching_sets(bs, be)
    t1 = T()
    bsh, beh = bs.cpu().numpy(), be.cpu().numpy()
    index_of = {id(s): i for i, s in enumerate(pl.sets)}
    orientation = rules.choose_barcoding_kit([pl.sets[i] for i in matching], lambda s: bsh[index_of[id(s)]], lambda s: beh[index_of[id(s)]])
    bc_sets = [i for i in matching if rules.is_barcode(pl.sets[i]) and rules.barcode_direction(pl.sets[i]) == orientation]
    names, bins = barcode_bins(pl, bc_sets)
    t2 = T()
    jobs, where = pl._phase_b_jobs(reads, matching)
    _, out, rec_off = pl._scan_jobs(reads.arena, jobs, MODE_TRACE, p.end_size, with_layout=True)
    t3 = T()
    if rep == 2:
        # where the scan's non-kernel time goes
        ta = T(); woff = torch.cat([j[1] for j in jobs[::2]]); wlen = torch.cat([j[2] for j in jobs[::2]]).to(torch.int32); tb = T()
        o2 = torch.empty((len(jobs) * n, 8), dtype=torch.int32, device="cuda"); tc = T()
        import numpy as np
        starts = np.arange(len(jobs) // 2 + 1, dtype=np.int64) * n
        ja = np.array([j[0] for j in jobs[::2]], dtype=np.int32); jb_ = np.array([j[0] for j in jobs[1::2]], dtype=np.int32)
        h0 = time.perf_counter()
        pl.aligner.scan_device(reads.arena, woff, wlen, ja, starts, p.end_size, o2, MODE_TRACE, job_adapter_b=jb_)
        h1 = time.perf_counter(); td = T()
        pl.aligner.scan_device(reads.arena, woff, wlen, ja, starts, p.end_size, o2, MODE_TRACE, job_adapter_b=jb_)
        h2 = time.perf_counter(); te = T()
        print("   cat %.1f ms | empty %.1f | scan_device host %.1f + wait %.1f (first, new job table) | again: host %.1f + wait %.1f" %
              ((tb - ta) * 1e3, (tc - tb) * 1e3, (h1 - h0) * 1e3, (td - h1) * 1e3, (h2 - td) * 1e3, (te - h2) * 1e3))
        pl.aligner.get_timing()
        del woff, wlen, o2
    st = torch.zeros(n, dtype=torch.int32, device="cuda"); et = torch.zeros_like(st); call = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    job_of = {(si, side): k for k, (side, si) in enumerate(where)}
    jb = [(job_of.get((b[0], 0), -1), job_of.get((b[1], 1), -1)) for b in bins]
    pl.aligner.phase_b_reduce(out, n, rec_off, [w[0] for w in where], p.end_size, p.min_trim_size, p.extra_end_trim, p.end_threshold, st, et,
                              bins=jb, barcode_threshold=75.0, barcode_diff=5.0, require_two=False, call=call)
    t4 = T()
    c = call.to(torch.int64).cpu().numpy()
    t5 = T()
    tm = pl.aligner.get_timing()
    print("rep %d: phase A %.1f ms | kit %.1f | phase-B scan %.1f (kernels %.1f) | reduce %.1f | calls to host %.1f | total %.1f ms" %
          (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, tm["trace"][0], (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
