"""Worker for bench.py's cpu_baseline leg (TEST INFRASTRUCTURE): runs the reference's sequential
per-read logic (tests/ref_pipeline.py) over a chunk of reads in a fresh process that never touches
the GPU, through the compiled reference (oracle/_ref) when present, else the oracle port."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


class _Set:
    def __init__(self, name, start, end):
        self.name, self.start, self.end = name, start, end


class _Params:
    def __init__(self, d):
        self.__dict__.update(d)


def _backend(want_ref):
    from oracle.oracle import Oracle, Reference, REF_SO
    return Reference() if (want_ref and os.path.isfile(REF_SO)) else Oracle()


def run_chunk(args):
    """Phases B + C of a chunk of reads -> (reads done, seconds, per-read results): one
    (start_trim, end_trim, [(adapter, start, end), ...]) per read, so that the caller can compare
    them with what the GPU produced for the same reads."""
    seqs, sets, matching, params, want_ref = args[:5]
    middle = args[5] if len(args) > 5 else True
    from tests import ref_pipeline
    fn = _backend(want_ref).adapter_alignment
    sets = [_Set(*s) for s in sets]
    p = _Params(params)
    adapters = []
    for i in matching:
        s = sets[i]
        if s.start is not None:
            adapters.append(s.start)
        if s.end is not None and (s.start is None or s.end[1] != s.start[1]):
            adapters.append(s.end)
    out = []
    t0 = time.perf_counter()
    for seq in seqs:
        st, et = ref_pipeline.phase_b(fn, seq, sets, matching, p)
        hits = ref_pipeline.phase_c(fn, seq, st, et, adapters, p) if middle else []
        out.append((st, et, [(a, rs, re) for a, rs, re, _ in hits]))
    return len(seqs), time.perf_counter() - t0, out


def run_chunk_barcodes(args):
    """Phase B with barcode scores + the barcode call of a chunk of reads (demultiplexing run)
    -> (reads done, seconds, [(start_trim, end_trim, bin name), ...])."""
    seqs, sets, matching, params, want_ref, orientation, thr, diff, two = args
    from tests import ref_pipeline
    fn = _backend(want_ref).adapter_alignment
    sets = [_Set(*s) for s in sets]
    p = _Params(params)
    out = []
    t0 = time.perf_counter()
    for seq in seqs:
        st, et, ss, es = ref_pipeline.phase_b_barcodes(fn, seq, sets, matching, p, orientation)
        out.append((st, et, ref_pipeline.determine_barcode(ss, es, thr, diff, two)))
    return len(seqs), time.perf_counter() - t0, out


def run_chunk_phase_a(args):
    """Phase A (adapter-set presence, nanopore_read.py:149-164) of a chunk of reads on the CPU: the best start / end
    full-adapter identity of every set over these reads -> (reads done, seconds, (best_start, best_end))."""
    seqs, sets, params, want_ref = args
    from tests import ref_pipeline
    fn = _backend(want_ref).adapter_alignment
    sets = [_Set(*s) for s in sets]
    p = _Params(params)
    t0 = time.perf_counter()
    bs, be = ref_pipeline.phase_a(fn, seqs, sets, p)
    return len(seqs), time.perf_counter() - t0, (bs, be)


def run_chunk_demux_middle(args):
    """A whole demultiplexing run with the middle scan on (BASELINE configs[4] shape) for a chunk of reads: phase B with
    barcode scores, the barcode call, phase C over every matching set's sequences
    -> (reads done, seconds, [(start_trim, end_trim, bin name, [(adapter, start, end), ...]), ...])."""
    seqs, sets, matching, params, want_ref, orientation, thr, diff, two = args
    from tests import ref_pipeline
    fn = _backend(want_ref).adapter_alignment
    sets = [_Set(*s) for s in sets]
    p = _Params(params)
    adapters = []
    for i in matching:
        s = sets[i]
        if s.start is not None:
            adapters.append(s.start)
        if s.end is not None and (s.start is None or s.end[1] != s.start[1]):
            adapters.append(s.end)
    out = []
    t0 = time.perf_counter()
    for seq in seqs:
        st, et, ss, es = ref_pipeline.phase_b_barcodes(fn, seq, sets, matching, p, orientation)
        call = ref_pipeline.determine_barcode(ss, es, thr, diff, two)
        hits = ref_pipeline.phase_c(fn, seq, st, et, adapters, p)
        out.append((st, et, call, [(a, rs, re) for a, rs, re, _ in hits]))
    return len(seqs), time.perf_counter() - t0, out
