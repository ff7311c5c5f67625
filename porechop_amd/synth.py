"""Synthetic read sets of SURVEY.md section 8d / BASELINE.md section 3, generated on the device.

Reads are exactly `read_len` bases of i.i.d. uniform ACGT; a fraction carries a mutated (10 %,
sub:ins:del = 4:3:3) and possibly outer-truncated (30 % of copies, U{0..20} bases) copy of the
start adapter at its first bases / of the end adapter at its last bases; a fraction
(`chimera_frac`) carries an end+start adapter junction (5 % mutation) at a uniform position in
[read_len/8, 7*read_len/8].  Adapter copies come from a pool of `pool` instances drawn with
Python's `random.Random(seed)`; bodies and assignments from a seeded torch generator.  (The
survey's generator draws every base with random.Random; at 8 Gbase that is hours of Python, so
only the adapter instances are drawn that way.  Copies OVERWRITE the body's first/last bases, so
every read stays exactly `read_len` long.)
"""
import random

import numpy as np
import torch

from .pipeline import DeviceReads

Y_TOP = "AATGTACTTCGTTCAGTTACGTATTGCT"       # SQK-NSK007_Y_Top    (porechop/adapters.py:78)
Y_BOTTOM = "GCAATACGTAACTGAACGAAGT"          # SQK-NSK007_Y_Bottom (porechop/adapters.py:79)


def _mutate(rng, seq, rate):
    out = []
    for c in seq:
        x = rng.random()
        if x < rate * 0.4:
            out.append(rng.choice("ACGT"))
        elif x < rate * 0.7:
            out.append(c)
            out.append(rng.choice("ACGT"))
        elif x < rate:
            pass
        else:
            out.append(c)
    return "".join(out)


def _pool(rng, seq, n, rate, trunc_side, width):
    """n mutated (and, for adapters, sometimes outer-truncated) copies of seq, left-aligned in [n, width]."""
    arr = np.full((n, width), ord("A"), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int64)
    for i in range(n):
        s = _mutate(rng, seq, rate)
        if trunc_side and rng.random() < 0.3:
            k = rng.randint(0, 20)
            s = s[k:] if trunc_side == "front" else s[:max(0, len(s) - k)]
        s = s[:width]
        lens[i] = len(s)
        arr[i, :len(s)] = np.frombuffer(s.encode(), dtype=np.uint8)
    return arr, lens


def make_reads(n_reads, read_len=8000, seed=1, start_frac=0.9, end_frac=0.5, chimera_frac=0.0,
               start_adapter=Y_TOP, end_adapter=Y_BOTTOM, device="cuda", pool=4096,
               barcodes_start=None, barcodes_end=None, barcode_rate=0.10):
    """barcodes_start / barcodes_end: lists of equally many
    sequences (the panel's BCb / BCb_rev, porechop/adapters.py:176-463).  Every read then draws
    b ~ U{0..B-1} and carries a mutated copy of barcodes_start[b] between the start adapter (if it
    has one) and the body, and of barcodes_end[b] between the body and the end adapter;
    the drawn b is returned as `reads.truth_barcode` (int64 [n_reads])."""
    dev = torch.device(device)
    rng = random.Random(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = n_reads * read_len
    arena = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    chunk = 1 << 26
    for s in range(0, total, chunk):
        e = min(total, s + chunk)
        r = torch.randint(0, 4, (e - s,), dtype=torch.uint8, device=dev, generator=g)
        arena[s:e] = acgt[r.long()]
    arena[total:] = ord("N")
    view = arena[:total].view(n_reads, read_len)

    def instances(seq, rate, trunc_side, count=pool):
        width = len(seq) + 8
        inst, lens = _pool(rng, seq, count, rate, trunc_side, width)
        return torch.from_numpy(inst).to(dev), torch.from_numpy(lens).to(dev), width

    def paste(sel, inst_rows, L, dist, width, region, at_end):
        """Write instance bytes into the first (at_end=False) / last (True) `region` columns of the
        reads `sel`: instance row i occupies columns [dist_i, dist_i + L_i) counted from the read's
        start, or -- at_end -- ends dist_i bases before the read's last base."""
        if sel.numel() == 0:
            return
        c = torch.arange(region, device=dev)
        rows = view[sel, read_len - region:] if at_end else view[sel, :region]
        first = (region - dist - L) if at_end else dist
        src = c[None, :] - first[:, None]
        m = (src >= 0) & (src < L[:, None])
        vals = torch.gather(inst_rows, 1, torch.clamp(src, min=0, max=width - 1))
        rows[m] = vals[m]
        if at_end:
            view[sel, read_len - region:] = rows
        else:
            view[sel, :region] = rows

    truth = None
    for at_end, frac, seq, trunc, bcs in ((False, start_frac, start_adapter, "front", barcodes_start),
                                          (True, end_frac, end_adapter, "back", barcodes_end)):
        has_ad = torch.zeros(n_reads, dtype=torch.bool, device=dev)
        La = torch.zeros(n_reads, dtype=torch.int64, device=dev)
        ad = None
        if frac > 0 and seq is not None:
            inst, lens, width = instances(seq, 0.10, trunc)
            has_ad = torch.rand(n_reads, device=dev, generator=g) < frac
            k = torch.randint(0, pool, (n_reads,), device=dev, generator=g)
            La = torch.where(has_ad, lens[k], La)
            ad = (inst, lens, width, k)
        if bcs:
            nb = len(bcs)
            per = 64                                        # mutated instances per barcode
            if truth is None:
                truth = torch.randint(0, nb, (n_reads,), device=dev, generator=g)
            wb = max(len(b) for b in bcs) + 8
            binst = torch.full((nb * per, wb), ord("A"), dtype=torch.uint8, device=dev)
            blens = torch.zeros(nb * per, dtype=torch.int64, device=dev)
            for b, bseq in enumerate(bcs):
                i_, l_, w_ = instances(bseq, barcode_rate, None, per)
                binst[b * per:(b + 1) * per, :w_] = i_
                blens[b * per:(b + 1) * per] = l_
            kb = truth * per + torch.randint(0, per, (n_reads,), device=dev, generator=g)
            region = wb + (ad[2] if ad is not None else 0)
            for s0 in range(0, n_reads, 1 << 18):           # bounded temporaries
                sel = torch.arange(s0, min(n_reads, s0 + (1 << 18)), device=dev)
                paste(sel, binst[kb[sel]], blens[kb[sel]], La[sel], wb, region, at_end)
        if ad is not None:
            inst, lens, width, k = ad
            sel = torch.nonzero(has_ad).flatten()
            for s0 in range(0, int(sel.numel()), 1 << 18):
                ss = sel[s0:s0 + (1 << 18)]
                paste(ss, inst[k[ss]], lens[k[ss]], torch.zeros_like(ss), width, width, at_end)
    if chimera_frac > 0 and start_adapter and end_adapter:
        junction = end_adapter + start_adapter
        inst, lens, width = instances(junction, 0.05, None)
        sel = torch.nonzero(torch.rand(n_reads, device=dev, generator=g) < chimera_frac).flatten()
        k = torch.randint(0, pool, (sel.numel(),), device=dev, generator=g)
        lo, hi = read_len // 8, 7 * read_len // 8 - width
        pos = torch.randint(lo, max(lo + 1, hi), (sel.numel(),), device=dev, generator=g)
        c = torch.arange(width, device=dev)
        idx = pos[:, None] + c[None, :]
        rows = torch.gather(view[sel], 1, idx)
        m = c[None, :] < lens[k][:, None]
        rows[m] = inst[k][m]
        tmp = view[sel]
        tmp.scatter_(1, idx, rows)
        view[sel] = tmp
    off = torch.arange(n_reads, device=dev, dtype=torch.int64) * read_len
    length = torch.full((n_reads,), read_len, dtype=torch.int32, device=dev)
    reads = DeviceReads(arena, off, length)
    reads.truth_barcode = truth
    return reads


def make_ragged_reads(n_reads, mean_len=8000, sigma=0.6, min_len=20, seed=1, start_frac=0.9, end_frac=0.5, chimera_frac=0.0,
                      start_adapter=Y_TOP, end_adapter=Y_BOTTOM, device="cuda", pool=4096):
    """Like make_reads, but with log-normal read lengths (mean mean_len, shape sigma, floor min_len): the
    length distribution of a real nanopore run instead of exactly read_len bases per read.  Reads lie
    back to back in one arena; adapter copies overwrite the first / last bases of reads long enough to
    hold them, junctions go to a uniform position inside reads of at least 2 000 bases."""
    dev = torch.device(device)
    rng = random.Random(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    z = torch.randn(n_reads, device=dev, generator=g, dtype=torch.float64)
    length = torch.clamp(torch.round(mean_len * torch.exp(sigma * z - 0.5 * sigma * sigma)), min=min_len).to(torch.int64)
    off = torch.cumsum(length, 0) - length
    total = int(length.sum().item())
    arena = torch.empty(total + 64, dtype=torch.uint8, device=dev)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    chunk = 1 << 26
    for s0 in range(0, total, chunk):
        e0 = min(total, s0 + chunk)
        arena[s0:e0] = acgt[torch.randint(0, 4, (e0 - s0,), dtype=torch.uint8, device=dev, generator=g).long()]
    arena[total:] = ord("N")

    def paste(frac, seq, rate, trunc, where):
        if frac <= 0 or not seq:
            return
        width = len(seq) + 8
        inst, lens = _pool(rng, seq, pool, rate, trunc, width)
        inst, lens = torch.from_numpy(inst).to(dev), torch.from_numpy(lens).to(dev)
        need = width if where != "middle" else 2000
        sel = torch.nonzero((torch.rand(n_reads, device=dev, generator=g) < frac) & (length >= need)).flatten()
        if sel.numel() == 0:
            return
        k = torch.randint(0, pool, (sel.numel(),), device=dev, generator=g)
        L = lens[k]
        if where == "start":
            first = off[sel]
        elif where == "end":
            first = off[sel] + length[sel] - L
        else:
            u = torch.rand(sel.numel(), device=dev, generator=g, dtype=torch.float64)
            lo = length[sel] // 8
            hi = torch.clamp(7 * length[sel] // 8 - width, min=1)
            first = off[sel] + lo + (u * torch.clamp(hi - lo, min=1).to(torch.float64)).to(torch.int64)
        c = torch.arange(width, device=dev)
        m = c[None, :] < L[:, None]
        idx = first[:, None] + c[None, :]
        arena[idx[m]] = inst[k][m]

    paste(start_frac, start_adapter, 0.10, "front", "start")
    paste(end_frac, end_adapter, 0.10, "back", "end")
    if chimera_frac > 0 and start_adapter and end_adapter:
        paste(chimera_frac, end_adapter + start_adapter, 0.05, None, "middle")
    return DeviceReads(arena, off, length.to(torch.int32))


def reads_from_strings(seqs, device="cuda"):
    """Upload Python strings the way NanoporeRead.__init__ normalises them
    (porechop/nanopore_read.py:26-31): upper-case, and U->T when U's outnumber T's."""
    norm = []
    for s in seqs:
        s = s.upper()
        if s.count("U") > s.count("T"):
            s = s.replace("U", "T")
        norm.append(s)
    lens = np.array([len(s) for s in norm], dtype=np.int32)
    offs = np.zeros(len(norm), dtype=np.int64)
    if len(norm) > 1:
        offs[1:] = np.cumsum(lens[:-1].astype(np.int64))
    blob = "".join(norm).encode() + b"N" * 64
    dev = torch.device(device)
    arena = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(dev)
    return DeviceReads(arena, torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)), norm
