"""Ahead-of-time build of the specialised score kernels of the static adapter panel.

The whole-read score pass runs a kernel specialised for one adapter pair (csrc/pc_jit.cpp).  The panel is a
constant (porechop/adapters.py:77-463), and the batch pipeline scans the start and end sequence of one adapter set
in one pass (Pipeline._scan_jobs pairs the jobs of a set), so the kernels a run needs are known when the library is
built: one per set.  `python -m porechop_amd.aot` (called by __graft_entry__.build() and `make kernels`) compiles
them with hiprtc -- no GPU needed -- into porechop_amd/kernel_cache/, which the library consults before it ever
compiles at run time.  Anything else (custom adapters, other scoring schemes, leftover singles paired with each
other) is compiled once at run time and kept in the user's cache directory (PC_JIT_CACHE_DIR / ~/.cache/porechop_amd).
"""
import os
import sys
from typing import List, Optional, Tuple

DEFAULT_SCORES = (3, -6, -5, -2)            # porechop/porechop.py:145
HERE = os.path.dirname(os.path.abspath(__file__))
CACHE_DIR = os.path.join(HERE, "kernel_cache")


def canonical_pair(a: str, b: Optional[str]) -> Tuple[str, Optional[str]]:
    """The order in which Pipeline._scan_jobs hands two adapters that share their windows to the library: the
    longer one first, the given order on ties.  (The kernel is specialised for the ordered pair.)"""
    if b is None or b == a:
        return a, None
    return (a, b) if len(a) >= len(b) else (b, a)


def panel_kernel_pairs(panel, full_native=range(1, 13), full_rapid=range(1, 13)) -> List[Tuple[str, Optional[str]]]:
    """The (ordered) adapter pairs whose kernels ship with the library: every set of the panel -- its start and end
    sequence together, or its one sequence alone -- every sequence of the panel alone as well, plus the full native barcode adapters 1-12 (the panel has 12 reverse barcodes) and the full rapid
    barcode adapters 1-12 (porechop.py:410-436; higher numbers are compiled at first use)."""
    from . import panel as rules
    pairs, seen = [], set()

    def add(s):
        seqs = [x[1] for x in (s.start, s.end) if x is not None]
        if not seqs:
            return
        p = canonical_pair(seqs[0], seqs[1] if len(seqs) > 1 else None)
        if p not in seen:
            seen.add(p)
            pairs.append(p)

    for s in panel:
        add(s)
    # every sequence of the panel alone: the score pass of the pruned phase B scans end windows one adapter per kernel
    for s in panel:
        for x in (s.start, s.end):
            if x is not None and (x[1], None) not in seen:
                seen.add((x[1], None))
                pairs.append((x[1], None))
    # the barcode pairs of the pruned phase B's score pass (panel.phase_b_pair_key): start with start, end with end
    by_key = {}
    for s in panel:
        k = rules.phase_b_pair_key(s)
        if k is not None:
            by_key.setdefault(k, []).append(s)
    for group in by_key.values():
        if len(group) == 2:
            for side in ("start", "end"):
                x, y = getattr(group[0], side), getattr(group[1], side)
                if x is not None and y is not None and x[1] != y[1]:
                    p = canonical_pair(x[1], y[1])
                    if p not in seen:
                        seen.add(p)
                        pairs.append(p)
    for i in full_native:
        add(rules.full_native_barcode(panel, i))
    for i in full_rapid:
        add(rules.full_rapid_barcode_old(panel, i))
        add(rules.full_rapid_barcode_new(panel, i))
    return pairs


def _build_one(args):
    a, b, scores, cache_dir = args
    from ._lib import load_library
    lib = load_library()
    return lib.pc_jit_precompile(a.encode(), (b or "").encode(), *scores, cache_dir.encode())


def prebuild(cache_dir: str = CACHE_DIR, scores=DEFAULT_SCORES, workers: Optional[int] = None, pairs=None, quiet=False):
    """Compile every kernel of panel_kernel_pairs() that is not in cache_dir yet, `workers` processes at a time.
    -> (compiled, already there, failed)"""
    import multiprocessing as mp
    from .panel import load_panel
    if pairs is None:
        pairs = panel_kernel_pairs(load_panel(prefer_reference=False))
    os.makedirs(cache_dir, exist_ok=True)
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0))))
    jobs = [(a, b, tuple(scores), cache_dir) for a, b in pairs]
    if workers == 1:
        res = [_build_one(j) for j in jobs]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_build_one, jobs, chunksize=4)
    done, there, failed = sum(r == 0 for r in res), sum(r == 1 for r in res), sum(r < 0 for r in res)
    if not quiet:
        print("porechop_amd.aot: %d kernels compiled, %d already in %s, %d failed" % (done, there, cache_dir, failed))
    return done, there, failed


if __name__ == "__main__":
    d, t, f = prebuild(sys.argv[1] if len(sys.argv) > 1 else CACHE_DIR)
    sys.exit(1 if f else 0)
