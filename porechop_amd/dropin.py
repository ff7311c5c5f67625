"""Drop the GPU core under an UNCHANGED Porechop: batch what its three phase drivers are about
to ask, one pair at a time, through ``adapter_alignment``.

    import porechop.porechop as pp          # the reference package, unmodified
    import porechop_amd.dropin as dropin
    dropin.install(pp)
    pp.main()

``install`` replaces ``porechop.nanopore_read.adapter_alignment`` (the name ``align_adapter``
resolves at call time, porechop/nanopore_read.py:476-477) by a memoising function and wraps

    find_matching_adapter_sets      porechop/porechop.py:286-327   (phase A)
    find_adapters_at_read_ends      porechop/porechop.py:438-514   (phase B)
    find_adapters_in_read_middles   porechop/porechop.py:533-595   (phase C)

so that, before the original function runs, every (window, adapter) pair it will request has
been aligned in a few large GPU batches and sits in the memo.  Phase C's mask-and-realign loop
(nanopore_read.py:210-243) is replayed here on the memoised results to discover which further
alignments the reference will ask for; those are batched round by round.  The original
functions then run untouched and only ever hit the memo (``stats()['misses'] == 0``).

The backend is any object with ``align(pairs, scores) -> list[str]`` (pairs = [(read, adapter)]);
the default is the GPU library.  There is no CPU backend in this package; tests inject the oracle.
"""
import functools
import itertools
import time


class GpuBackend:
    """(read, adapter) string pairs -> the reference's 7-field strings, through the C ABI (pc_align_batch_host +
    pc_format_results).  Every distinct read string goes into the arena once, however many adapters it meets; one
    context per scoring scheme is kept for the life of the backend."""

    def __init__(self):
        self._aligners = {}

    def _aligner(self, adapters, scores):
        from .batch import Aligner
        al = self._aligners.get(scores)
        if al is None:
            al = self._aligners[scores] = Aligner(adapters, scores)
        else:
            al.set_adapters(adapters)
        return al

    def align(self, pairs, scores):
        import numpy as np
        from .batch import format_results
        if not pairs:
            return []
        ads, ad_idx, rd_idx = [], {}, {}
        chunks, pos = [], 0
        n = len(pairs)
        offs = np.empty(n, dtype=np.int64)
        lens = np.empty(n, dtype=np.int32)
        idx = np.empty(n, dtype=np.int32)
        for p, (rd, ad) in enumerate(pairs):
            ai = ad_idx.get(ad)
            if ai is None:
                ai = ad_idx[ad] = len(ads)
                ads.append(ad)
            where = rd_idx.get(rd)
            if where is None:
                b = rd.encode()
                where = rd_idx[rd] = (pos, len(b))
                chunks.append(b)
                pos += len(b)
            offs[p], lens[p] = where
            idx[p] = ai
        al = self._aligner(ads, tuple(int(x) for x in scores))
        return format_results(al.align_host(b"".join(chunks), offs, lens, idx))

    def align_product(self, reads, adapters, scores):
        """Every read against every adapter (distinct strings each) -> the strings in read-major order.  No per-pair Python:
        the arena holds each read once, the pair table is numpy repeat / tile."""
        import numpy as np
        from .batch import format_results
        if not reads or not adapters:
            return []
        enc = [r.encode() for r in reads]
        lens1 = np.fromiter((len(b) for b in enc), dtype=np.int64, count=len(enc))
        offs1 = np.concatenate([[0], np.cumsum(lens1[:-1])]).astype(np.int64)
        A = len(adapters)
        al = self._aligner(list(adapters), tuple(int(x) for x in scores))
        recs = al.align_host(b"".join(enc), np.repeat(offs1, A), np.repeat(lens1, A).astype(np.int32),
                             np.tile(np.arange(A, dtype=np.int32), len(enc)))
        return format_results(recs)

    def close(self):
        for al in self._aligners.values():
            al.close()
        self._aligners = {}


class _State:
    def __init__(self, backend):
        self.backend = backend
        self.memo = {}
        self.hits = 0
        self.misses = 0
        self.batched = 0
        self.prefetch_seconds = 0.0       # wall clock inside prefetch(), of which ...
        self.backend_seconds = 0.0        # ... inside the backend's batch calls

    def prefetch(self, pairs, scores):
        t0 = time.perf_counter()
        try:
            self._prefetch(pairs, scores)
        finally:
            self.prefetch_seconds += time.perf_counter() - t0

    def _prefetch(self, pairs, scores):
        key_scores = tuple(scores)
        todo, seen = [], set()
        for rd, ad in pairs:
            k = (rd, ad, key_scores)
            if k not in self.memo and k not in seen:
                seen.add(k)
                todo.append((rd, ad))
        if not todo:
            return
        t0 = time.perf_counter()
        out = self.backend.align(todo, key_scores)
        self.backend_seconds += time.perf_counter() - t0
        self.batched += len(todo)
        for (rd, ad), res in zip(todo, out):
            self.memo[(rd, ad, key_scores)] = res

    def prefetch_product(self, reads, adapters, scores):
        """prefetch() of every read x every adapter -- the shape of phase A, phase B and round 0 of phase C -- without a
        Python-level loop over the pairs when the backend has align_product (the GPU backend does)."""
        if not hasattr(self.backend, "align_product"):
            return self.prefetch([(r, a) for r in reads for a in adapters], scores)
        t0 = time.perf_counter()
        try:
            ks = tuple(scores)
            adapters = list(dict.fromkeys(adapters))
            memo = self.memo
            # reads of which some pair is still unknown (a read is usually either wholly new or wholly known)
            reads = [r for r in dict.fromkeys(reads) if any((r, a, ks) not in memo for a in adapters)]
            if not reads or not adapters:
                return
            t1 = time.perf_counter()
            out = self.backend.align_product(reads, adapters, ks)
            self.backend_seconds += time.perf_counter() - t1
            self.batched += len(out)
            memo.update(zip(itertools.product(reads, adapters, (ks,)), out))
        finally:
            self.prefetch_seconds += time.perf_counter() - t0

    def lookup(self, read_sequence, adapter_sequence, scoring_scheme_vals):
        k = (read_sequence, adapter_sequence, tuple(scoring_scheme_vals))
        r = self.memo.get(k)
        if r is not None:
            self.hits += 1
            return r
        self.misses += 1
        r = self.backend.align([(read_sequence, adapter_sequence)], tuple(scoring_scheme_vals))[0]
        self.memo[k] = r
        return r


_state = None


def stats():
    return {"hits": _state.hits, "misses": _state.misses, "batched": _state.batched,
            "entries": len(_state.memo)} if _state else {}


def _parse(result):
    # porechop/nanopore_read.py:476-491 (align_adapter)
    parts = result.split(',')
    read_start = int(parts[0])
    if read_start == -1:
        return 0.0, 0.0, -1, 0
    return float(parts[6]), float(parts[5]), read_start, int(parts[1]) + 1


def install(pp, backend=None):
    """pp: the imported (unchanged) ``porechop.porechop`` module.  Returns the state object."""
    global _state
    import importlib
    nr = importlib.import_module(pp.__name__.rsplit('.', 1)[0] + '.nanopore_read')
    st = _State(backend if backend is not None else GpuBackend())
    _state = st
    nr.adapter_alignment = st.lookup

    told = []

    def note_threads(orig, args, kw):
        """The reference's --threads pool (porechop.py:309-322,484-509,575-591) has nothing left to parallelise here -- every
        alignment is a memo lookup -- and its threads only hand the GIL to one another: measured, --threads 16 is 40 % SLOWER
        than --threads 1 under this drop-in.  Said once, on stderr; nothing is changed."""
        if told:
            return
        try:
            import inspect
            threads = inspect.signature(orig).bind(*args, **kw).arguments.get("threads", 1)
        except Exception:
            return
        if isinstance(threads, int) and threads > 1:
            import sys
            told.append(1)
            print("porechop_amd.dropin: --threads %d only adds hand-offs of Python's GIL around memo lookups (the alignments "
                  "are batched onto the GPU before the reference's loops run); --threads 1 is faster here" % threads, file=sys.stderr)

    orig_a = pp.find_matching_adapter_sets
    orig_b = pp.find_adapters_at_read_ends
    orig_c = pp.find_adapters_in_read_middles

    @functools.wraps(orig_a)
    def find_matching_adapter_sets(check_reads, verbosity, end_size, scoring_scheme_vals, *args, **kw):
        note_threads(orig_a, (check_reads, verbosity, end_size, scoring_scheme_vals) + args, kw)
        search = [a for a in pp.ADAPTERS if '(full sequence)' not in a.name]      # porechop.py:296
        # nanopore_read.py:155,160: every check read's start window x every start sequence, end window x every end sequence
        st.prefetch_product([read.seq[:end_size] for read in check_reads],
                            [s.start_sequence[1] for s in search if s.start_sequence], scoring_scheme_vals)
        st.prefetch_product([read.seq[-end_size:] for read in check_reads],
                            [s.end_sequence[1] for s in search if s.end_sequence], scoring_scheme_vals)
        return orig_a(check_reads, verbosity, end_size, scoring_scheme_vals, *args, **kw)

    @functools.wraps(orig_b)
    def find_adapters_at_read_ends(reads, matching_sets, verbosity, end_size, extra_trim_size, end_threshold,
                                   scoring_scheme_vals, *args, **kw):
        # nanopore_read.py:174,196
        st.prefetch_product([read.seq[:end_size] for read in reads],
                            [s.start_sequence[1] for s in matching_sets if s.start_sequence], scoring_scheme_vals)
        st.prefetch_product([read.seq[-end_size:] for read in reads],
                            [s.end_sequence[1] for s in matching_sets if s.end_sequence], scoring_scheme_vals)
        return orig_b(reads, matching_sets, verbosity, end_size, extra_trim_size, end_threshold,
                      scoring_scheme_vals, *args, **kw)

    @functools.wraps(orig_c)
    def find_adapters_in_read_middles(reads, matching_sets, verbosity, middle_threshold, extra_trim_good_side,
                                      extra_trim_bad_side, scoring_scheme_vals, *args, **kw):
        adapters = []                                                             # porechop.py:541-548
        for ms in matching_sets:
            if ms.start_sequence:
                adapters.append(ms.start_sequence[1])
            if ms.end_sequence:
                if (not ms.start_sequence) or ms.end_sequence[1] != ms.start_sequence[1]:
                    adapters.append(ms.end_sequence[1])
        key_scores = tuple(scoring_scheme_vals)
        seqs = [r.get_seq_with_start_end_adapters_trimmed() for r in reads]       # nanopore_read.py:216
        # round 0: every adapter against every unmasked trimmed read
        st.prefetch_product(seqs, adapters, scoring_scheme_vals)
        # replay nanopore_read.py:217-243 on the memo to learn what else will be asked
        active = []                       # [masked_seq, adapter index] of reads still in the loop
        for s in seqs:
            active.append([s, 0])
        while active:
            need, nxt = [], []
            for item in active:
                masked, ai = item
                while ai < len(adapters):
                    res = st.memo.get((masked, adapters[ai], key_scores))
                    if res is None:
                        need.append((masked, adapters[ai]))
                        break
                    full, _, rs, re = _parse(res)
                    if full >= middle_threshold:
                        masked = masked[:rs] + '-' * (re - rs) + masked[re:]
                    else:
                        ai += 1
                item[0], item[1] = masked, ai
                if ai < len(adapters):
                    nxt.append(item)
            st.prefetch(need, scoring_scheme_vals)
            active = nxt
        return orig_c(reads, matching_sets, verbosity, middle_threshold, extra_trim_good_side,
                      extra_trim_bad_side, scoring_scheme_vals, *args, **kw)

    pp.find_matching_adapter_sets = find_matching_adapter_sets
    pp.find_adapters_at_read_ends = find_adapters_at_read_ends
    pp.find_adapters_in_read_middles = find_adapters_in_read_middles
    return st
