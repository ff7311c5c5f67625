import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import load_panel_sets, load_panel_json
from porechop_amd.batch import MODE_SCORE, MODE_TRACE
from porechop_amd.pipeline import Pipeline, ScanParams
from porechop_amd.synth import make_reads
p = ScanParams(); pl = Pipeline(load_panel_sets(), p)
fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
reads = make_reads(6000, 8000, seed=12, start_frac=0.9, end_frac=0.5, barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
matching = [i for i, s in enumerate(pl.sets) if s.name == "SQK-NSK007" or (s.name.startswith("Barcode ") and "(forward)" in s.name)]
jobs, where = pl._phase_b_jobs(reads, matching)
so, sl = pl._end_windows(reads, None, "start"); eo, el = pl._end_windows(reads, None, "end")
score = torch.stack(pl._scan_jobs(reads.arena, jobs, MODE_SCORE, p.end_size, fuse=False))
ub, ubf = pl._phase_b_bounds(score, jobs, where, sl, el)
is_end = torch.tensor([w[0] for w in where], dtype=torch.bool, device=ub.device)
print("odd", int((score[..., 0] != -2).sum()), "of", score[..., 0].numel())
for side in (0, 1):
    u = ub[is_end == bool(side)]
    print("side", side, "ub quantiles", [int(torch.quantile(u.float().flatten()[:2000000], q)) for q in (0.1, 0.5, 0.9, 0.99)], "frac ub>=26", float((u >= 26).float().mean()), "frac ub>=56", float((u >= 56).float().mean()))
    sc = score[is_end == bool(side)]
    idx = torch.nonzero(u >= 26)[:12]
    for j, r in idx.tolist():
        print("   ", sc[j, r].tolist()[:5], "ub", int(u[j, r]))

for side, thr in ((0, 56), (0, 28), (1, 50), (1, 28)):
    rows = torch.nonzero(is_end == bool(side)).flatten()
    u = ub[rows]; sc = score[rows]
    print("side", side, "frac ub>0 %.3f  frac ub>%d %.3f" % (float((u > 0).float().mean()), thr, float((u > thr).float().mean())))
    idx = torch.nonzero(u > thr)
    pick = idx[torch.randperm(idx.shape[0], device=idx.device)[:10]]
    for j, r in pick.tolist():
        print("    score rec (J, I, S) =", sc[j, r].tolist()[1:5:1], "ub", int(u[j, r]))
