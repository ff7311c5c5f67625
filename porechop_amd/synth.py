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
               start_adapter=Y_TOP, end_adapter=Y_BOTTOM, device="cuda", pool=4096):
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
    col = None

    def paste(mask_frac, seq, rate, trunc_side, at_end):
        nonlocal col
        if mask_frac <= 0 or seq is None:
            return
        width = len(seq) + 8
        inst, lens = _pool(rng, seq, pool, rate, trunc_side, width)
        inst = torch.from_numpy(inst).to(dev)
        lens = torch.from_numpy(lens).to(dev)
        sel = torch.nonzero(torch.rand(n_reads, device=dev, generator=g) < mask_frac).flatten()
        k = torch.randint(0, pool, (sel.numel(),), device=dev, generator=g)
        L = lens[k]
        c = torch.arange(width, device=dev)
        if not at_end:
            rows = view[sel, :width]
            m = c[None, :] < L[:, None]
            rows[m] = inst[k][m]
            view[sel, :width] = rows
        else:
            rows = view[sel, read_len - width:]
            # right-aligned: instance byte i goes to column width - L + i
            src = c[None, :] - (width - L)[:, None]
            m = src >= 0
            vals = torch.gather(inst[k], 1, torch.clamp(src, min=0))
            rows[m] = vals[m]
            view[sel, read_len - width:] = rows

    paste(start_frac, start_adapter, 0.10, "front", False)
    paste(end_frac, end_adapter, 0.10, "back", True)
    if chimera_frac > 0 and start_adapter and end_adapter:
        junction = end_adapter + start_adapter
        width = len(junction) + 8
        inst, lens = _pool(rng, junction, pool, 0.05, None, width)
        inst = torch.from_numpy(inst).to(dev)
        lens = torch.from_numpy(lens).to(dev)
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
    return DeviceReads(arena, off, length)


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
