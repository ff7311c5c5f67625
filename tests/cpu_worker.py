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


def run_chunk(args):
    seqs, sets, matching, params, want_ref = args
    from oracle.oracle import Oracle, Reference, REF_SO
    from tests import ref_pipeline
    backend = Reference() if (want_ref and os.path.isfile(REF_SO)) else Oracle()
    fn = backend.adapter_alignment
    sets = [_Set(*s) for s in sets]
    p = _Params(params)
    adapters = []
    for i in matching:
        s = sets[i]
        if s.start is not None:
            adapters.append(s.start)
        if s.end is not None and (s.start is None or s.end[1] != s.start[1]):
            adapters.append(s.end)
    t0 = time.perf_counter()
    for seq in seqs:
        st, et = ref_pipeline.phase_b(fn, seq, sets, matching, p)
        ref_pipeline.phase_c(fn, seq, st, et, adapters, p)
    return len(seqs), time.perf_counter() - t0
