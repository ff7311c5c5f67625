"""Batched form of Porechop's three adapter-search phases, driving the GPU library.

This is the host-side mirror of the CALLERS of the hot path, batched so that the GPU sees
millions of (window, adapter) pairs per launch instead of one pair per ctypes call:

  phase A  porechop/porechop.py:286-327   find_matching_adapter_sets (+ nanopore_read.py:149-164)
  phase B  porechop/porechop.py:438-514   find_adapters_at_read_ends (+ nanopore_read.py:166-208)
  phase C  porechop/porechop.py:533-595   find_adapters_in_read_middles (+ nanopore_read.py:210-243)

Every decision (thresholds, trim arithmetic, the sequential mask-and-realign loop of phase C) is
the reference's, evaluated with torch tensor ops on the device from the integer records the
kernels return; the alignments themselves only ever come from the C ABI (no CPU path).
PyTorch is plumbing here: it owns the HBM buffers and the stream.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import os

import numpy as np
import torch

from .batch import MODE_SCORE, MODE_TRACE, MODE_TRACE_AT, MODE_TWO_PASS, Aligner, RESULT_INTS

# phase B is pruned (exactly: see "Exact pruning of phase B" below) from this many (sequence, side) jobs on; measured on
# MI355X, 1 M reads: 198 jobs 192 -> 122 ms; with the 4-6 jobs of a run without barcodes tracing everything is faster
PRUNE_MIN_JOBS = int(os.environ.get("PC_PRUNE_MIN_JOBS", "24"))
# up to this many adapter sets the prefilter stage of the middle scan takes ONE host round trip (survivors counted per set on
# the device); a barcode panel's ~100 sets keep the single compaction over all of them
LEAN_PREFILTER_GROUPS = 8


@dataclass
class AdapterSet:
    """Mirror of porechop/adapters.py:18-52 (name + optional (name, seq) start / end)."""
    name: str
    start: Optional[Tuple[str, str]] = None
    end: Optional[Tuple[str, str]] = None


@dataclass
class ScanParams:
    # defaults of porechop/porechop.py:138-185
    end_size: int = 150
    min_trim_size: int = 4
    extra_end_trim: int = 2
    end_threshold: float = 75.0
    middle_threshold: float = 90.0
    adapter_threshold: float = 90.0
    check_reads: int = 10000
    scores: Tuple[int, int, int, int] = (3, -6, -5, -2)


@dataclass
class DeviceReads:
    """Reads resident in HBM: one byte per base as delivered (upper-cased ASCII), read r at
    arena[off[r] : off[r] + length[r]].  The arena must have >= 16 readable bytes after the last
    read (the kernels fetch bases a dword at a time)."""
    arena: Optional[torch.Tensor]    # uint8 [bytes]; None = the reads are held PACKED only (plane / exc / ends below)
    off: torch.Tensor      # int64 [R]
    length: torch.Tensor   # int32 [R]
    # reads held at 2 bits per base (packed_only): the plane io.pack_reads makes (base i of the arena in bits 2 (i % 16) of
    # dword i / 16; 64 readable bytes past the last base), the ascending positions of the bases that were not A/C/G/T/U, and
    # byte copies of every read's two end windows: (arena, start window offsets, end window offsets, end_size)
    plane: Optional[torch.Tensor] = None
    exc: Optional[torch.Tensor] = None
    nbases: int = 0
    ends: Optional[tuple] = None

    @property
    def n(self):
        return int(self.off.shape[0])

    @classmethod
    def packed_only(cls, aligner, packed, nbases, exceptions, off, length, end_size=150):
        """Reads that crossed PCIe at 2 bits per base and STAY packed in HBM (a quarter of the bytes): only what the DP kernels
        read is ever turned into bytes -- here the two end windows of every read (pc_unpack_windows: 2 x end_size bytes per
        read), later the few whole reads that survive the exact prefilter, which itself scans the plane
        (pc_prefilter_packed).  off / length index the plane in BASES (the byte offsets of the arena it was packed from)."""
        dev = packed.device
        need = (int(nbases) + 15) // 16 * 4 + 64
        if int(packed.numel()) < need:                         # the scan fetches whole 16-byte blocks: slack past the last base
            grown = torch.zeros(need, dtype=torch.uint8, device=dev)
            grown[:packed.numel()] = packed
            packed = grown
        n = int(off.shape[0])
        ln64 = length.to(torch.int64)
        wl = torch.clamp(length, max=end_size).to(torch.int32).contiguous()
        stride = (end_size + 15) // 16 * 16 + 16
        dst_off = torch.arange(2 * n + 1, dtype=torch.int64, device=dev) * stride
        ends = torch.empty(2 * n * stride + 64, dtype=torch.uint8, device=dev)
        ends[2 * n * stride:] = ord("N")
        if n:
            aligner.unpack_windows(packed, exceptions, off.contiguous(), wl, ends, dst_off[:n + 1].contiguous())
            aligner.unpack_windows(packed, exceptions, (off + (ln64 - wl.to(torch.int64))).contiguous(), wl, ends, dst_off[n:].contiguous())
        return cls(None, off, length, plane=packed, exc=exceptions, nbases=int(nbases),
                   ends=(ends, dst_off[:n], dst_off[n:2 * n], int(end_size)))

    def materialize(self, aligner):
        """The byte arena of packed-only reads (pc_unpack_device), for the routes that need every base as a byte."""
        if self.arena is None:
            self.arena = aligner.unpack_device(self.plane[:(self.nbases + 15) // 16 * 4], self.nbases, self.exc)
        return self.arena

    @classmethod
    def from_packed(cls, aligner, packed, nbases, exceptions, off, length, arena=None):
        """Reads that crossed PCIe at 2 bits per base (io.pack_reads) -> the resident byte arena (pc_unpack_device:
        'A' / 'C' / 'G' / 'T', 'N' at the listed exceptions, 64 bytes of 'N' behind the last read)."""
        return cls(aligner.unpack_device(packed, nbases, exceptions, arena=arena), off, length)


@dataclass
class MiddleHits:
    read: torch.Tensor       # int64 [H]   read index
    adapter: torch.Tensor    # int32 [H]   index into Pipeline.middle_adapters
    start: torch.Tensor      # int32 [H]   read_start in trimmed-read coordinates
    end: torch.Tensor        # int32 [H]   read_end (exclusive)
    identity: torch.Tensor   # float64 [H] full-adapter identity
    rounds: int = 0
    alignments: int = 0


def _round6(x):
    """float("%f" % x) for non-negative doubles: the value Python parses back from the C side's six printed decimals.
    The quotient is a TRUE division by a one-element device tensor: `tensor / 1e6` with a Python scalar is evaluated by
    PyTorch as a multiplication by the rounded reciprocal, which lands one ulp off the correctly rounded value for about
    one identity in seven (seen as 33 of 238 presence-table entries differing from the host's in the last bit)."""
    million = torch.full((1,), 1e6, dtype=torch.float64, device=x.device)
    return torch.round(x * 1e6) / million


def _identities(rec):
    """float64 (full, partial) exactly as nanopore_read.py:476-491 parses them: the C side prints
    (100.0*matches)/len with %f and Python float()s it back; rounding the unrounded double to six decimals
    reproduces the printed digits as long as no value lies within 5e-7 of a rounding boundary without being on
    it -- identities are ratios of small integers (len <= a few hundred), whose six-decimal digits are never
    within 1e-8 of a tie."""
    m = rec[..., 5].to(torch.float64)
    partial = _round6(100.0 * m / rec[..., 6].to(torch.float64))
    full = _round6(100.0 * m / rec[..., 7].to(torch.float64))
    return full, partial


def trimmed_interval(length, start_trim, end_trim):
    """[s, e) of seq[start_trim : len(seq) - end_trim] with Python's slice semantics, the whole read
    when both trims are 0 (nanopore_read.py:56-62): a start beyond the end clamps, a negative end
    index counts from the end.  Tensors in, int64 tensors out (e < s means empty)."""
    ln = length.to(torch.int64)
    s_pos = torch.clamp(start_trim.to(torch.int64), max=ln)
    e_pos = ln - end_trim.to(torch.int64)
    e_pos = torch.where(e_pos < 0, torch.clamp(ln + e_pos, min=0), e_pos)
    untouched = (start_trim == 0) & (end_trim == 0)
    s_pos = torch.where(untouched, torch.zeros_like(s_pos), s_pos)
    e_pos = torch.where(untouched, ln, e_pos)
    return s_pos, e_pos


def call_barcodes(nbins: int, start_scores: torch.Tensor, end_scores: torch.Tensor, barcode_threshold: float,
                  barcode_diff: float, require_two_barcodes: bool) -> np.ndarray:
    """nanopore_read.py:399-466 for every read at once (torch formulation; the GPU product path runs
    the same rules in pc_phase_b_reduce).

    start_scores / end_scores are float64 [R, K]: the full-adapter identity of bin k's start / end
    sequence (the reference's two dicts, in insertion order; bin names are distinct).
    -> int64 [R] bin index, or -1 for 'none'.

    Ties are resolved as Python's stable sorted(..., reverse=True) resolves them there: among equal
    scores the entry inserted first wins, start entries before end entries."""
    R, K = start_scores.shape
    dev = start_scores.device
    none = torch.full((R,), -1, dtype=torch.int64, device=dev)
    if K == 0:
        return none.cpu().numpy()       # best = ('none', 0.0): the call is 'none' whatever the thresholds

    def best_two(x):
        order = torch.sort(x, dim=1, descending=True, stable=True)
        second = order.values[:, 1] if x.shape[1] >= 2 else torch.zeros(R, dtype=x.dtype, device=dev)
        return order.indices[:, 0], order.values[:, 0], second

    if require_two_barcodes:
        si, sv, s2 = best_two(start_scores)
        ei, ev, e2 = best_two(end_scores)
        ok = (sv >= barcode_threshold) & (ev >= barcode_threshold) & \
             (sv >= s2 + barcode_diff) & (ev >= e2 + barcode_diff)
        ok &= si == ei                                             # start_end_match compares NAMES (distinct per bin)
        call = torch.where(ok, si, none)
    else:
        both = torch.cat([start_scores, end_scores], dim=1)            # start entries first
        bi, bv, _ = best_two(both)
        bk = bi % K
        # second best = best score among the OTHER names (each name keeps its best of start/end)
        per_name = torch.maximum(start_scores, end_scores)
        other = per_name.masked_fill(torch.arange(K, device=dev)[None, :] == bk[:, None], -1.0)
        second = torch.clamp(other.max(dim=1).values, min=0.0)
        ok = (bv >= barcode_threshold) & (bv >= second + barcode_diff)
        call = torch.where(ok, bk, none)
    return call.cpu().numpy()


class Pipeline:
    def __init__(self, sets: List[AdapterSet], params: ScanParams = None, device=None, aligner=None):
        """aligner: an object with the Aligner interface.  The default -- and the only product
        path -- is the GPU library; tests inject an oracle-backed stand-in to check the host logic
        on machines without a GPU."""
        self.sets = []
        self.p = params or ScanParams()
        self.seq_index = {}
        self.seqs = []
        self._register(sets)
        if aligner is None:
            self.device = torch.device(device if device is not None else "cuda")
            dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            self.aligner = Aligner(self.seqs, self.p.scores, device=dev_index)
        else:
            self.device = torch.device(device if device is not None else "cpu")
            self.aligner = aligner
            self.aligner.set_adapters(self.seqs)
        self.stats = {"pairs_end": 0, "pairs_middle": 0, "cells_end": 0, "cells_middle": 0}
        self._consts = {}

    def _const(self, values, dtype=torch.int64):
        """A small read-only device tensor of host values (job -> set / side / adapter length tables ...), kept between calls: a
        torch.tensor(list, device=cuda) is a pageable upload that WAITS for everything enqueued before it -- half a dozen of
        them per step were half a dozen drains of the GPU's queue (0.5 ms each at 1 M reads), for tables that repeat from batch
        to batch."""
        key = (dtype, tuple(values))
        t = self._consts.get(key)
        if t is None:
            if len(self._consts) > 512:
                self._consts.clear()
            t = self._consts[key] = torch.tensor(list(values), dtype=dtype, device=self.device)
        return t

    def packed_kernels(self):
        """True when every pair of this pipeline runs the packed 16-bit kernels (pc_scores_supported): the exact prunings
        below (score bounds, PC_MODE_SCORE records) are derived for those schemes.  Any other scheme -- the reference takes
        any four integers -- and adapters above 128 bases run the library's plain-int32 kernel: same answers, every record
        computed, no pruning."""
        lib = getattr(self.aligner, "lib", None)
        if lib is None or not hasattr(lib, "pc_scores_supported"):
            # test stand-ins (no library): the sign conditions the bounds' derivations rest on
            match, mismatch, go, ge = [int(x) for x in self.p.scores]
            return match > 0 and match > mismatch and go < 0 and ge < 0
        longest = max([len(x) for x in self.seqs] + [1])
        return bool(lib.pc_scores_supported(*[int(x) for x in self.p.scores], int(longest)))

    def _register(self, sets):
        # one adapter table for everything (deduplicated sequences)
        for s in sets:
            self.sets.append(s)
            for side in (s.start, s.end):
                if side is not None and side[1] not in self.seq_index:
                    self.seq_index[side[1]] = len(self.seqs)
                    self.seqs.append(side[1])

    def add_sets(self, sets: List[AdapterSet]) -> List[int]:
        """Append adapter sets found necessary after phase A (porechop.py:410-436: the full barcode
        adapters) -> their indices in self.sets."""
        first = len(self.sets)
        self._register(sets)
        self.aligner.set_adapters(self.seqs)
        return list(range(first, len(self.sets)))

    # ------------------------------------------------------------------------------------------
    def _end_windows(self, reads: DeviceReads, idx: Optional[torch.Tensor], side: str):
        """seq[:end_size] / seq[-end_size:] as (offset, length) windows (nanopore_read.py:155,160)."""
        ln = reads.length if idx is None else reads.length[idx]
        wl = torch.clamp(ln, max=self.p.end_size)
        if reads.arena is None:                      # packed-only reads: the windows' byte copies (DeviceReads.packed_only)
            assert reads.ends is not None and reads.ends[3] == self.p.end_size
            tab = reads.ends[1] if side == "start" else reads.ends[2]
            return (tab if idx is None else tab[idx]), wl
        off = reads.off if idx is None else reads.off[idx]
        if side == "start":
            return off, wl
        return off + (ln - wl).to(torch.int64), wl

    @staticmethod
    def _ends_arena(reads: DeviceReads):
        """The bytes the end windows index: the read arena, or -- packed-only reads -- the copies of the end windows."""
        return reads.arena if reads.arena is not None else reads.ends[0]

    def _scan_jobs(self, arena, jobs, mode, max_len, with_layout=False, sort_lengths=False, typ_len=0, fuse=True):
        """jobs: list of (adapter_index, win_off int64[n], win_len int32[n]) -> list of [n,8] views.

        Jobs that scan the very same windows (same tensors) are fused two adapters at a time
        (similar lengths together), so each window is streamed from HBM once per adapter PAIR.

        sort_lengths: whole-read scans of reads of different lengths.  A tile (64 or 128 consecutive
        windows) runs as many columns as its longest window, so the windows of every job are handed
        over longest first -- a tile then holds windows of nearly one length, and the specialised score
        kernel stays on its block-resolved path -- and the records are put back in the caller's order.
        typ_len: their typical (mean) length, which lets the library cut the longest reads into chunks of
        about that many columns (pc_set_length_hint) so that a launch does not last as long as its longest read."""
        order_of = {}
        if sort_lengths:
            sorted_jobs = []
            for j in jobs:
                key = (id(j[1]), id(j[2]))
                if key not in order_of:
                    order = torch.argsort(j[2], descending=True, stable=True)
                    order_of[key] = (order, j[1][order], j[2][order], j[1], j[2])   # keeps the originals alive (ids stay unique)
                sorted_jobs.append((j[0], order_of[key][1], order_of[key][2]) + tuple(j[3:]))
            orders = [order_of[(id(j[1]), id(j[2]))][0] for j in jobs]
            jobs = sorted_jobs
        if os.environ.get("PC_NO_FUSE", "0") not in ("", "0"):        # (timing experiments: every adapter alone, exact rows, two read streams)
            fuse = False
        groups = {}
        for k, j in enumerate(jobs):
            # fuse=False: every job alone (one adapter per kernel: the single-sequence kernels ship with the library)
            groups.setdefault((id(j[1]), id(j[2])) if fuse else k, []).append(k)
        fused = []                      # (job index a, job index b or None)
        for ks in groups.values():
            # jobs that carry the same pairing hint (the two sequences of one adapter set, phase C) go together,
            # longer adapter first: a set's pair is the same in every run, so its specialised score kernel can
            # be -- and for the static panel is -- built ahead of time (porechop_amd/aot.py)
            by_hint, rest = {}, []
            for k in ks:
                h = jobs[k][3] if len(jobs[k]) > 3 else None
                if h is None:
                    rest.append(k)
                else:
                    by_hint.setdefault(h, []).append(k)
            for h, hk in by_hint.items():
                if isinstance(h, tuple) and h and h[0] == "alone":
                    fused.extend((k, None) for k in hk)           # scanned with its single-sequence kernel
                elif len(hk) == 2 and self.seqs[jobs[hk[0]][0]] != self.seqs[jobs[hk[1]][0]]:
                    a, b = hk
                    fused.append((a, b) if len(self.seqs[jobs[a][0]]) >= len(self.seqs[jobs[b][0]]) else (b, a))
                else:
                    rest.extend(hk)
            ks = sorted(rest, key=lambda k: -len(self.seqs[jobs[k][0]]))
            for i in range(0, len(ks) - 1, 2):
                fused.append((ks[i], ks[i + 1]))
            if len(ks) % 2:
                fused.append((ks[-1], None))
        woff = torch.cat([jobs[a][1] for a, _ in fused])
        wlen = torch.cat([jobs[a][2] for a, _ in fused]).to(torch.int32)
        starts = np.zeros(len(fused) + 1, dtype=np.int64)
        ostarts = np.zeros(len(fused) + 1, dtype=np.int64)
        for k, (a, b) in enumerate(fused):
            n = jobs[a][1].shape[0]
            starts[k + 1] = starts[k] + n
            ostarts[k + 1] = ostarts[k] + n * (2 if b is not None else 1)
        out = torch.empty((int(ostarts[-1]), RESULT_INTS), dtype=torch.int32, device=self.device)
        if starts[-1] > 0:
            self.aligner.set_length_hint(typ_len if sort_lengths else 0)
            self.aligner.scan_device(arena, woff, wlen, np.array([jobs[a][0] for a, _ in fused], dtype=np.int32),
                                     starts, max_len, out, mode,
                                     job_adapter_b=np.array([jobs[b][0] if b is not None else -1 for _, b in fused],
                                                            dtype=np.int32))
        res = [None] * len(jobs)
        rec_off = [0] * len(jobs)
        for k, (a, b) in enumerate(fused):
            n = jobs[a][1].shape[0]
            o = int(ostarts[k])
            res[a] = out[o:o + n]; rec_off[a] = o
            if b is not None:
                res[b] = out[o + n:o + 2 * n]; rec_off[b] = o + n
        if sort_lengths:
            assert not with_layout
            back = []
            for r, order in zip(res, orders):
                u = torch.empty_like(r)
                u[order] = r
                back.append(u)
            return back
        if with_layout:
            return res, out, rec_off          # + the one tensor behind the views and each job's first record
        return res

    # ------------------------------------------------------------------------------------------
    def presence_score_bound(self, m: int) -> Optional[int]:
        return self.identity_score_bound(m, self.p.adapter_threshold)

    def identity_score_bound(self, m: int, threshold: float) -> Optional[int]:
        """Smallest raw score an alignment of an m-base adapter can have if its full-adapter identity
        reaches `threshold` percent (None when the scheme gives no bound).  With t = threshold/100:
        identity >= t means matches M >= t * L_full >= t * m and non-match columns N <= M (1-t)/t;
        every scored non-match column costs at most P = max(|mismatch|, |gap_open|, |gap_extend|) and
        overhanging adapter bases cost nothing, so score >= match*M - P*N >= m * (t*match - P*(1-t))."""
        match, mismatch, go, ge = self.p.scores
        t = (threshold - 1e-6) / 100.0                          # the identity is compared after %f rounding
        per_base = t * match - max(-mismatch, -go, -ge) * (1.0 - t)
        if per_base <= 0:
            return None
        return int(np.floor(m * per_base))

    def phase_a(self, reads: DeviceReads, check_idx: Optional[torch.Tensor] = None, prune: bool = False):
        """-> (best_start[S], best_end[S]) float64 on device: the max full-adapter identity of every
        set's start / end sequence over the check reads.  This table is the only cross-read
        reduction in Porechop (nanopore_read.py:159,164); a multi-GPU run all-reduces it (MAX).

        prune=True: a score-only pass first, then traceback only for the pairs whose
        score can still mean an identity >= --adapter_threshold (presence_score_bound).  Which sets
        reach the threshold, and their best identities, are exactly the unpruned ones; the table
        entries of sets that do NOT reach it become lower bounds (they only feed a display in the
        reference), which is why this is an option and not the default."""
        S = len(self.sets)
        best_start = torch.zeros(S, dtype=torch.float64, device=self.device)
        best_end = torch.zeros(S, dtype=torch.float64, device=self.device)
        n = reads.n if check_idx is None else int(check_idx.shape[0])
        if n == 0:
            return best_start, best_end
        so, sl = self._end_windows(reads, check_idx, "start")
        eo, el = self._end_windows(reads, check_idx, "end")
        jobs, where = [], []
        for si, s in enumerate(self.sets):
            if "(full sequence)" in s.name:       # porechop.py:296
                continue
            if s.start is not None:
                jobs.append((self.seq_index[s.start[1]], so, sl)); where.append((si, 0))
            if s.end is not None:
                jobs.append((self.seq_index[s.end[1]], eo, el)); where.append((si, 1))
        if prune and self.packed_kernels():
            return self._phase_a_pruned(reads, jobs, where, best_start, best_end)
        outs = self._scan_jobs(self._ends_arena(reads), jobs, MODE_TRACE, self.p.end_size)
        # one vectorised reduction for all jobs (they all cover the same n check reads)
        rec = torch.stack(outs)                                     # [J, n, 8]
        m = rec[:, :, 5].to(torch.float64)
        full = _round6(100.0 * m / rec[:, :, 7].to(torch.float64))
        full = torch.where(rec[:, :, 0] == -1, torch.zeros_like(full), full).amax(dim=1)   # [J]
        # (which job feeds which entry is known on the host: index lists, not boolean masks -- a mask is a nonzero, a round trip)
        for side, best in ((0, best_start), (1, best_end)):
            ks = [k for k, w in enumerate(where) if w[1] == side]
            if ks:
                best.scatter_reduce_(0, self._const([where[k][0] for k in ks]), full[self._const(ks)], reduce="amax")
        self.stats["pairs_end"] += sum(int(j[1].shape[0]) for j in jobs)
        return best_start, best_end

    def _phase_a_pruned(self, reads, jobs, where, best_start, best_end):
        scores = torch.stack(self._scan_jobs(self._ends_arena(reads), jobs, MODE_SCORE, self.p.end_size))[:, :, 4]   # [J, n]
        bounds = [self.presence_score_bound(len(self.seqs[j[0]])) for j in jobs]
        need = self._const([b if b is not None else -(1 << 30) for b in bounds])
        cand = scores >= need[:, None]
        hit = torch.nonzero(cand)                              # [C, 2] (job, window), job-major: one sync
        counts = torch.bincount(hit[:, 0], minlength=len(jobs)).cpu().numpy()
        cjobs, cwhere = [], []
        pos = 0
        for k, (j, w) in enumerate(zip(jobs, where)):
            if counts[k]:
                sel = hit[pos:pos + int(counts[k]), 1]
                pos += int(counts[k])
                cjobs.append((j[0], j[1][sel], j[2][sel])); cwhere.append(w)
        self.stats["pairs_end"] += sum(int(j[1].shape[0]) for j in jobs)
        self.stats["pairs_end_traced_after_pruning"] = self.stats.get("pairs_end_traced_after_pruning", 0) + int(counts.sum())
        if not cjobs:
            return best_start, best_end
        outs = self._scan_jobs(self._ends_arena(reads), cjobs, MODE_TRACE, self.p.end_size)
        rec = torch.cat(outs)                                   # one reduction for all candidate jobs
        full, _ = _identities(rec)
        full = torch.where(rec[:, 0] == -1, torch.zeros_like(full), full)
        job_of = torch.repeat_interleave(torch.arange(len(cjobs), device=self.device),
                                         torch.tensor([int(j[1].shape[0]) for j in cjobs], device=self.device))
        top = torch.zeros(len(cjobs), dtype=torch.float64, device=self.device).scatter_reduce_(0, job_of, full, reduce="amax")
        for side, best in ((0, best_start), (1, best_end)):
            ks = [k for k, w in enumerate(cwhere) if w[1] == side]
            if ks:
                best.scatter_reduce_(0, self._const([cwhere[k][0] for k in ks]), top[self._const(ks)], reduce="amax")
        return best_start, best_end

    def matching_sets(self, best_start, best_end):
        """porechop.py:327: sets whose best start-or-end score reaches --adapter_threshold."""
        best = torch.maximum(best_start, best_end).cpu().numpy()
        return [i for i, s in enumerate(self.sets)
                if "(full sequence)" not in s.name and best[i] >= self.p.adapter_threshold]

    # ------------------------------------------------------------------------------------------
    def _phase_b_jobs(self, reads, matching):
        so, sl = self._end_windows(reads, None, "start")
        eo, el = self._end_windows(reads, None, "end")
        jobs, where = [], []
        for si in matching:
            s = self.sets[si]
            if s.start is not None:
                jobs.append((self.seq_index[s.start[1]], so, sl)); where.append((0, si))
            if s.end is not None:
                jobs.append((self.seq_index[s.end[1]], eo, el)); where.append((1, si))
        return jobs, where

    @property
    def native_reduce(self):
        """The per-read reduction of phase B runs as a library kernel (pc_phase_b_reduce) on the GPU;
        an injected test aligner without it takes the equivalent torch formulation below."""
        return hasattr(self.aligner, "phase_b_reduce")

    # ------------------------------------------------------------------------------------------
    # Exact pruning of phase B.  Of the ~200 end-window alignments a barcoded read gets, three matter: phase B keeps the
    # MAXIMUM trim over the alignments that qualify (nanopore_read.py:166-208) and the barcode call looks at full
    # identities within --barcode_diff of the best (nanopore_read.py:399-466).  A score-only pass (5 instead of 13.25
    # packed ops per two cells, no trace slab) gives every alignment's end cell (I adapter bases and Jc window columns
    # consumed) and score S -- the reference's own end cell, the one the two-pass scan retraces -- and from those,
    # UPPER BOUNDS on what the alignment can contribute:
    #   full identity <= 100 min(I, Jc, m) / m              (matches <= diagonal columns <= bases consumed; span >= m)
    #   start window:  read_end <= Jc + 1, so trim <= Jc + 1 + extra; none if Jc + 1 < min_trim_size; none if the path ends
    #                  in the window's last column of a full window (then read_end == end_size, which disqualifies)
    #   end window:    a qualifying alignment starts on the first row at a column j0 >= 1 (read_start = j0; a path from the
    #                  first column has read_start 0, which disqualifies), consumes all I adapter bases and b = Jc - j0 window
    #                  columns with  S <= (match + g) b - g I  and  b <= I + (match I - S) / g,  g = min(|open|, |extend|);
    #                  so none if Jc - 1 < ceil((S + g I) / (match + g)), else trim <= end_size - max(1, Jc - bmax) + extra.
    #   either side:   a qualifying alignment has aligned identity > --end_threshold over Lp >= min(I, Jc) columns, hence
    #                  S > (tau match - (1 - tau) P) min(I, Jc)   (P = the dearest non-matching column)
    # Round 1 traces, per read and side, the two best-scoring pairs that can trim at all and the two best-scoring barcode
    # pairs; the exact reduction of those gives each read's trims so far and the best barcode identity per side; round 2 traces
    # the pairs whose trim bound still exceeds the trims, and the barcode pairs that could reach
    # max(best, --barcode_threshold) - --barcode_diff (a full identity of t also needs S >= m (t match - (1 - t) P)).
    # Every pair left untraced is proven unable to change the maximum
    # or the call and enters the reduction as "no alignment".  tests/test_gpu_phase_b_pruning.py checks the bounds against
    # the full records of EVERY pair of its batches and the results against the unpruned phase B.
    def _phase_b_bounds(self, score_rec, jobs, where, sl, el):
        """score_rec [J, R, 8] (MODE_SCORE records) -> (ub_trim int64 [J, R], ub_full float64 [J, R])."""
        p = self.p
        dev = self.device
        match, _, go, ge = p.scores
        g = min(-go, -ge)
        m = self._const([len(self.seqs[j[0]]) for j in jobs])[:, None]
        is_end = self._const([bool(w[0]) for w in where], torch.bool)[:, None]
        n = torch.where(is_end, el[None, :].to(torch.int64), sl[None, :].to(torch.int64))
        flag = score_rec[:, :, 0].to(torch.int64)
        Jc = score_rec[:, :, 1].to(torch.int64)
        I = score_rec[:, :, 2].to(torch.int64)
        S = score_rec[:, :, 4].to(torch.int64)
        ub_full = 100.0 * torch.minimum(torch.minimum(I, Jc), m).to(torch.float64) / m.to(torch.float64)
        # start windows
        ok_s = (Jc + 1 >= p.min_trim_size) & ~((Jc == n) & (n == p.end_size))
        ub_s = torch.where(ok_s, Jc + 1 + p.extra_end_trim, torch.zeros_like(Jc))
        # end windows
        bmin = torch.div(S + g * I + (match + g - 1), match + g, rounding_mode="floor")
        bmax = I + torch.div(torch.clamp(match * I - S, min=0), g, rounding_mode="floor")
        ok_e = (Jc - 1 >= bmin) & (bmax + 1 >= p.min_trim_size)
        ub_e = torch.where(ok_e, p.end_size - torch.clamp(Jc - bmax, min=1) + p.extra_end_trim, torch.zeros_like(Jc))
        ub = torch.where(is_end, ub_e, ub_s)
        # either side: a trim needs aligned identity > --end_threshold over the path's Lp columns, Lp >= min(I, Jc) (a path
        # from the first row consumes all I adapter bases, one from the first column all Jc window columns); with
        # M > tau Lp matches and fewer than (1 - tau) Lp other columns at a cost of at most P each,
        # S > (tau match - (1 - tau) P) Lp.  (Random 24-28-mers align into a full window at ~64 % identity with S = 7..17:
        # this is what prunes them.)
        tau = (p.end_threshold - 1e-6) / 100.0
        P = max(-p.scores[1], -go, -ge, 0)
        c = tau * match - (1.0 - tau) * P
        if c > 0:
            need_s = torch.floor(c * torch.clamp(torch.minimum(I, Jc), min=1).to(torch.float64)).to(torch.int64)
            ub = torch.where(S > need_s, ub, torch.zeros_like(ub))
        # and a full identity of t needs S >= m (t match - (1 - t) P)  (identity_score_bound, as in phase A's pruning)
        odd = flag != -2                                   # anything that is not a plain score record: trace it
        ub = torch.where(odd, torch.full_like(ub, 1 << 20), ub)
        ub_full = torch.where(odd, torch.full_like(ub_full, 100.0), ub_full)
        return ub, ub_full

    def _phase_b_pruned_records(self, reads, jobs, where, call_sets, call_level, reduce_trims, call_level_diff=0.0):
        """Records of phase B with only the pairs that can matter traced (see above); the others keep their score-only
        record, which the reduction reads as "no alignment".  call_sets: set indices whose full identities feed a barcode
        call (traced whenever they can reach call_level = threshold - diff, or come within call_level_diff of the best
        traced on their side).  reduce_trims(records, rec_off) -> (start_trim, end_trim) runs the exact reduction.
        The selection runs on the device (pc_select.hip); the host reads back one count per job and round.
        self.debug_bounds (tests): a dict that receives the bounds and the untouched score records.
        -> (records, rec_off); self._traced_mask = the pairs that hold a traced record (int64 [J, words])"""
        p = self.p
        dev = self.device
        al = self.aligner
        self._traced_mask = None
        bounds_out = getattr(self, "debug_bounds", None)
        R, J = reads.n, len(jobs)
        so, sl = self._end_windows(reads, None, "start")
        eo, el = self._end_windows(reads, None, "end")
        so, eo = so.contiguous(), eo.contiguous()
        sl, el = sl.to(torch.int32).contiguous(), el.to(torch.int32).contiguous()
        # barcodes 2k-1 and 2k of a direction share a pass over their windows (one read stream instead of two: 10 instead
        # of 8 TCUPS), every other sequence runs alone; both kinds of kernel ship with the library (porechop_amd/aot.py)
        from . import panel as rules
        hinted = []
        for k, (j, (side, si)) in enumerate(zip(jobs, where)):
            key = rules.phase_b_pair_key(self.sets[si])
            hinted.append(tuple(j[:3]) + ((("pair", side) + key) if key is not None else ("alone", k),))
        _, rec, rec_off = self._scan_jobs(self._ends_arena(reads), hinted, MODE_SCORE, p.end_size, with_layout=True)
        job_off = self._const(rec_off)
        job_side = self._const([w[0] for w in where], torch.int32)
        job_len = self._const([len(self.seqs[j[0]]) for j in jobs], torch.int32)
        job_calls = self._const([1 if w[1] in call_sets else 0 for w in where], torch.int32)
        job_adapter = np.array([j[0] for j in jobs], dtype=np.int32)
        words = (R + 63) // 64
        best_full = torch.zeros((2, R), dtype=torch.float64, device=dev)   # best traced full identity of a call pair, per side
        trace_at = getattr(al, "trace_at", False) and os.environ.get("PC_NO_TRACE_AT", "0") in ("", "0")

        def trace(mask, counts):
            cnt = counts.cpu().numpy()                                   # the round's one host round trip
            total = int(cnt.sum())
            if total == 0:
                return 0
            first = torch.cumsum(counts, 0) - counts
            woff = torch.empty(total, dtype=torch.int64, device=dev)
            wlen = torch.empty(total, dtype=torch.int32, device=dev)
            dest = torch.empty(total, dtype=torch.int64, device=dev)
            pjob = torch.empty(total, dtype=torch.int32, device=dev)
            pread = torch.empty(total, dtype=torch.int64, device=dev)
            cursor = torch.empty(J, dtype=torch.int64, device=dev)
            al.phase_b_gather(mask, R, first, cursor, job_off, job_side, so, sl, eo, el, woff, wlen, dest, pjob, pread)
            live = np.nonzero(cnt)[0]
            starts = np.zeros(len(live) + 1, dtype=np.int64)
            starts[1:] = np.cumsum(cnt[live])
            al.set_length_hint(0)
            if trace_at:
                # the end cell of every selected pair is known from its score record: only the columns its path can occupy
                # are traced (PC_MODE_TRACE_AT: the second pass of the whole-read scan, for end windows)
                traced = al.gather_records(rec, dest) if hasattr(al, "gather_records") else rec.index_select(0, dest)
                # (the library takes the pairs of a job by end column -- bucket_pairs -- so that a tile runs as many columns as
                # its LATEST end cell needs: adapters found at the start of their windows, J ~ 30 of 150, share tiles)
                al.scan_device(self._ends_arena(reads), woff, wlen, job_adapter[live], starts, p.end_size, traced, MODE_TRACE_AT)
            else:
                traced = torch.empty((total, RESULT_INTS), dtype=torch.int32, device=dev)
                al.scan_device(self._ends_arena(reads), woff, wlen, job_adapter[live], starts, p.end_size, traced, MODE_TRACE)
            al.phase_b_scatter(traced, dest, pjob, pread, rec, job_side, job_calls, best_full, R)
            return total

        def select(rnd, mask_prev=None, trims=(None, None)):
            mask = torch.empty((J, words), dtype=torch.int64, device=dev)
            counts = torch.empty(J, dtype=torch.int64, device=dev)
            ub_t = ub_f = None
            if rnd == 1 and bounds_out is not None:
                ub_t = torch.empty((J, R), dtype=torch.int32, device=dev)
                ub_f = torch.empty((J, R), dtype=torch.float64, device=dev)
                bounds_out.update(ub_trim=ub_t, ub_full=ub_f, score_records=rec.clone(), rec_off=list(rec_off))
            al.phase_b_select(rec, R, job_off, job_side, job_len, job_calls, sl, el, p.end_size, p.min_trim_size,
                              p.extra_end_trim, p.end_threshold, rnd, call_level, call_level_diff, mask, counts,
                              mask_prev=mask_prev, start_trim=trims[0], end_trim=trims[1], best_full=best_full,
                              ub_trim_out=ub_t, ub_full_out=ub_f)
            return mask, counts

        # round 1: per read and side the two best-SCORING pairs that can trim at all (the real adapter and the real barcode,
        # where the read has them) and the two best-scoring barcode pairs (they fix the level a rival would have to reach)
        mask1, counts1 = select(1)
        n1 = trace(mask1, counts1)
        st1, et1 = reduce_trims(rec, rec_off, mask1)
        # round 2: whatever could still beat the trims so far or change the call
        mask2, counts2 = select(2, mask_prev=mask1, trims=(st1, et1))
        n2 = trace(mask2, counts2)
        self.stats["pairs_end"] += J * R
        self.stats["pairs_end_traced_after_pruning"] = self.stats.get("pairs_end_traced_after_pruning", 0) + n1 + n2
        # which pairs hold a traced record: the reductions skip the others without loading them (pc_phase_b_reduce_masked)
        self._traced_mask = torch.bitwise_or(mask1, mask2)
        return rec, rec_off

    def _masked_kw(self, mask):
        """Keyword for phase_b_reduce: the traced-pairs mask, where the aligner's reduction takes one (the GPU library's does)."""
        if mask is None or not getattr(self.aligner, "reduce_takes_mask", False) or os.environ.get("PC_NO_REDUCE_MASK", "0") not in ("", "0"):
            return {}
        return {"traced_mask": mask.contiguous()}

    @property
    def can_prune_phase_b(self):
        """Needs the reduction and selection kernels and score records that carry the end cell (the GPU library's do)."""
        return self.native_reduce and hasattr(self.aligner, "phase_b_select") and getattr(self.aligner, "score_end_cell", False) and \
            self.packed_kernels()

    def _prune_b(self, prune, njobs):
        """prune=None: the exact pruning of phase B where it pays -- a barcode panel's worth of jobs (with a handful of
        adapters the score pass, two selections and two host round trips cost more than tracing everything)."""
        if prune is None:
            return self.can_prune_phase_b and njobs >= PRUNE_MIN_JOBS
        return bool(prune) and self.can_prune_phase_b

    def phase_b_demux(self, reads: DeviceReads, matching: List[int], bins, barcode_threshold, barcode_diff, require_two,
                      prune: Optional[bool] = None):
        """Phase B of a demultiplexing run: trims + the barcode call of every read.
        bins: one (start set index or None, end set index or None) per barcode bin, in the order the
        reference inserts the names into its score dicts (nanopore_read.py:185-187,206-208)
        -> (start_trim, end_trim int32[R], call: numpy int64[R], bin index or -1 = 'none')."""
        R = reads.n
        p = self.p
        if not self.native_reduce:
            full_for = {i for b in bins for i in b if i is not None}
            out = self.phase_b(reads, matching, full_for=full_for)
            fulls = out[2] if full_for else {}
            zeros = torch.zeros(R, dtype=torch.float64, device=self.device)
            S = [fulls.get((b[0], 0), zeros) if b[0] is not None else zeros for b in bins]
            E = [fulls.get((b[1], 1), zeros) if b[1] is not None else zeros for b in bins]
            S = torch.stack(S, dim=1) if bins else torch.zeros((R, 0), dtype=torch.float64, device=self.device)
            E = torch.stack(E, dim=1) if bins else torch.zeros((R, 0), dtype=torch.float64, device=self.device)
            return out[0], out[1], call_barcodes(len(bins), S, E, barcode_threshold, barcode_diff, require_two)
        start_trim = torch.zeros(R, dtype=torch.int32, device=self.device)
        end_trim = torch.zeros(R, dtype=torch.int32, device=self.device)
        call = torch.full((R,), -1, dtype=torch.int32, device=self.device)
        jobs, where = self._phase_b_jobs(reads, matching)
        masked = self._masked_kw
        tmask = None
        if jobs and R:
            sides = [w[0] for w in where]
            if self._prune_b(prune, len(jobs)):
                def trims(records, offs, mask=None):
                    a = torch.zeros(R, dtype=torch.int32, device=self.device)
                    b = torch.zeros(R, dtype=torch.int32, device=self.device)
                    self.aligner.phase_b_reduce(records, R, offs, sides, p.end_size, p.min_trim_size, p.extra_end_trim,
                                                p.end_threshold, a, b, **masked(mask))
                    return a, b
                call_sets = {i for b in bins for i in b if i is not None}
                out, rec_off = self._phase_b_pruned_records(reads, jobs, where, call_sets, barcode_threshold - barcode_diff, trims,
                                                            call_level_diff=barcode_diff)
                tmask = self._traced_mask
            else:
                _, out, rec_off = self._scan_jobs(self._ends_arena(reads), jobs, MODE_TRACE, p.end_size, with_layout=True)
                self.stats["pairs_end"] += sum(int(j[1].shape[0]) for j in jobs)
            job_of = {(si, side): k for k, (side, si) in enumerate(where)}
            jb = [(job_of.get((b[0], 0), -1) if b[0] is not None else -1,
                   job_of.get((b[1], 1), -1) if b[1] is not None else -1) for b in bins]
            self.aligner.phase_b_reduce(out, R, rec_off, sides, p.end_size, p.min_trim_size, p.extra_end_trim,
                                        p.end_threshold, start_trim, end_trim, bins=jb if bins else None,
                                        barcode_threshold=barcode_threshold, barcode_diff=barcode_diff,
                                        require_two=require_two, call=call, **masked(tmask))
        return start_trim, end_trim, call.to(torch.int64).cpu().numpy()

    def phase_b(self, reads: DeviceReads, matching: List[int], full_for=(), prune: Optional[bool] = None):
        """-> (start_trim[R], end_trim[R]) int32: nanopore_read.py:166-208 for every read.
        full_for: set indices whose full-adapter identities are wanted too (barcode calling,
        nanopore_read.py:185-187,206-208) -> third result {(set, side): float64[R]}, side 0 = start."""
        R = reads.n
        p = self.p
        start_trim = torch.zeros(R, dtype=torch.int32, device=self.device)
        end_trim = torch.zeros(R, dtype=torch.int32, device=self.device)
        jobs, where = self._phase_b_jobs(reads, matching)
        fulls = {}
        if not jobs:
            return (start_trim, end_trim, fulls) if full_for else (start_trim, end_trim)
        masked = self._masked_kw
        tmask = None
        if self.native_reduce and not full_for and R:
            sides = [w[0] for w in where]
            if self._prune_b(prune, len(jobs)):
                def trims(records, offs, mask=None):
                    a = torch.zeros(R, dtype=torch.int32, device=self.device)
                    b = torch.zeros(R, dtype=torch.int32, device=self.device)
                    self.aligner.phase_b_reduce(records, R, offs, sides, p.end_size, p.min_trim_size, p.extra_end_trim,
                                                p.end_threshold, a, b, **masked(mask))
                    return a, b
                out, rec_off = self._phase_b_pruned_records(reads, jobs, where, set(), 1e9, trims)
                tmask = self._traced_mask
            else:
                _, out, rec_off = self._scan_jobs(self._ends_arena(reads), jobs, MODE_TRACE, p.end_size, with_layout=True)
                self.stats["pairs_end"] += sum(int(j[1].shape[0]) for j in jobs)
            self.aligner.phase_b_reduce(out, R, rec_off, sides, p.end_size, p.min_trim_size, p.extra_end_trim,
                                        p.end_threshold, start_trim, end_trim, **masked(tmask))
            return start_trim, end_trim
        outs = self._scan_jobs(self._ends_arena(reads), jobs, MODE_TRACE, p.end_size)
        for (side, si), rec in zip(where, outs):
            full, partial = _identities(rec)
            ok = rec[:, 0] != -1
            if si in full_for:
                fulls[(si, side)] = torch.where(ok, full, torch.zeros_like(full))
            rs = rec[:, 0]
            re = rec[:, 1] + 1
            if side == 0:
                cond = ok & (partial > p.end_threshold) & (re != p.end_size) & ((re - rs) >= p.min_trim_size)
                start_trim = torch.where(cond, torch.maximum(start_trim, re + p.extra_end_trim), start_trim)
            else:
                cond = ok & (partial > p.end_threshold) & (rs != 0) & ((re - rs) >= p.min_trim_size)
                end_trim = torch.where(cond, torch.maximum(end_trim, (p.end_size - rs) + p.extra_end_trim), end_trim)
        self.stats["pairs_end"] += sum(int(j[1].shape[0]) for j in jobs)
        return (start_trim, end_trim, fulls) if full_for else (start_trim, end_trim)

    # ------------------------------------------------------------------------------------------
    def middle_adapter_list(self, matching: List[int]):
        """porechop.py:541-548: start sequence of every matching set, plus its end sequence when
        that differs from its own start sequence (duplicates across sets are kept, as there)."""
        return [a for a, _ in self._middle_adapters_with_sets(matching)]

    def _middle_adapters_with_sets(self, matching: List[int]):
        """middle_adapter_list with the index of the set each sequence comes from (the pairing hint of _scan_jobs)."""
        ads = []
        for si in matching:
            s = self.sets[si]
            if s.start is not None:
                ads.append((s.start, si))
            if s.end is not None and (s.start is None or s.end[1] != s.start[1]):
                ads.append((s.end, si))
        return ads

    def _prefiltered_scan(self, arena, off, length, max_len, a_list, aidx, hint, ks, ragged, typ_len, packed_reads=None):
        """Whole-read records of the adapters a_list (positions in the middle-adapter list) against the n windows
        (off, length), computed only where the exact prefilter cannot exclude a hit.  SPARSE result
        -> (b [C] int64: position in a_list, w [C] int64: window, rec [C, 8] int32); every (adapter, window) pair that
        is not listed is proven not to be a hit.  The windows that survive for one of a set's sequences are scanned for
        all of that set's sequences in one pass (the set's ahead-of-time kernel)."""
        dev = self.device
        n, B = int(off.shape[0]), len(a_list)
        e64 = torch.empty(0, dtype=torch.int64, device=dev)
        empty = (e64, e64, torch.empty((0, RESULT_INTS), dtype=torch.int32, device=dev))
        if n == 0 or B == 0:
            return empty
        order = None
        pf_off, pf_len = off.contiguous(), length.contiguous()
        if ragged:                                       # a wave runs 64 consecutive windows: similar lengths together
            order = torch.argsort(length, descending=True, stable=True)
            pf_off, pf_len = off[order].contiguous(), length[order].contiguous()
        self.aligner.set_length_hint(typ_len if ragged else 0)
        # packed_reads (reads held at 2 bits per base, `off` in bases): the prefilter scans the PLANE, and only the windows
        # that survive are turned into bytes for the DP; an adapter list the packed route does not take (a letter other than
        # A/C/G/T/U, a piece without seeds) falls back to unpacking everything
        groups, gidx = {}, []
        for a in a_list:
            gidx.append(groups.setdefault(hint[a], len(groups)))
        G = len(groups)
        members = [[b for b in range(B) if gidx[b] == g] for g in range(G)]
        al = self.aligner
        if (packed_reads is None or packed_reads.arena is not None) and G <= LEAN_PREFILTER_GROUPS and hasattr(al, "prefilter_defer_count") \
                and hasattr(torch, "nonzero_static") and os.environ.get("PC_NO_LEAN_PREFILTER", "0") in ("", "0"):
            # ONE host round trip for the whole stage (a handful of sets: the usual run).  The library's seed stage leaves its
            # candidate count on the device (pc_prefilter_defer_count), the survivors of every set are counted on the device,
            # the counts come back together, and the windows are listed by nonzero_static (no further synchronisation).
            adl, kl = [aidx[a] for a in a_list], [ks[a] for a in a_list]

            def survivors(defer):
                al.prefilter_defer_count(defer)
                try:
                    mask = al.prefilter_mask(arena, pf_off, pf_len, max_len, adl, kl)
                finally:
                    al.prefilter_defer_count(False)
                words = int(mask.shape[1])
                gm = []
                for g in range(G):
                    for wd in range(words):
                        bits = sum(1 << (b % 32) for b in members[g] if b // 32 == wd)
                        gm.append(bits - (1 << 32) if bits >= (1 << 31) else bits)
                if hasattr(al, "group_survivors") and os.environ.get("PC_NO_FUSED_GLUE", "0") in ("", "0"):
                    cand, cnt = al.group_survivors(mask, self._const(gm, torch.int32).view(G, words))
                    return [cand[g] for g in range(G)], cnt.cpu().numpy()               # the one synchronisation of this stage
                cands = []
                for g in range(G):
                    c = None
                    for wd in range(words):
                        if gm[g * words + wd]:
                            t = (mask[:, wd] & gm[g * words + wd]) != 0
                            c = t if c is None else (c | t)
                    cands.append(c)
                return cands, torch.stack([c.sum() for c in cands]).cpu().numpy()      # the one synchronisation of this stage
            cands, counts = survivors(True)
            if al.prefilter_overflowed():                # (rare: the seed list overflowed -- the mask above is incomplete)
                cands, counts = survivors(False)
            self.stats["pairs_middle_prefiltered"] = self.stats.get("pairs_middle_prefiltered", 0) + B * n
            cjobs, cmeta = [], []
            for g in range(G):
                if counts[g]:
                    sel = torch.nonzero_static(cands[g], size=int(counts[g])).flatten()
                    if order is not None:
                        sel = order[sel]
                    so, sl = off[sel], length[sel]
                    for b in members[g]:
                        cjobs.append((aidx[a_list[b]], so, sl, hint[a_list[b]])); cmeta.append((b, sel))
            if not cjobs:
                return empty
            outs = self._scan_jobs(arena, cjobs, MODE_TWO_PASS, max_len, sort_lengths=ragged, typ_len=typ_len)
            self.stats["pairs_middle_scanned_after_prefilter"] = self.stats.get("pairs_middle_scanned_after_prefilter", 0) + \
                sum(int(j[1].shape[0]) for j in cjobs)
            sb = torch.cat([torch.full((int(sel.shape[0]),), b, dtype=torch.int64, device=dev) for b, sel in cmeta])
            return sb, torch.cat([sel for _, sel in cmeta]), torch.cat(outs)
        got = None
        if packed_reads is not None and packed_reads.arena is None:
            got = self.aligner.prefilter_rows(packed_reads.plane, pf_off, pf_len, max_len, [aidx[a] for a in a_list], [ks[a] for a in a_list],
                                              packed=True)
            if got is None:
                arena = packed_reads.materialize(self.aligner)
                self.stats["packed_route_refused"] = self.stats.get("packed_route_refused", 0) + 1
        if got is None:
            got = self.aligner.prefilter_rows(arena, pf_off, pf_len, max_len, [aidx[a] for a in a_list], [ks[a] for a in a_list])
        rows, bits = got
        if order is not None:
            rows = order[rows]
        self.stats["pairs_middle_prefiltered"] = self.stats.get("pairs_middle_prefiltered", 0) + B * n
        if G == B:
            cand_g = bits
        else:                                            # union over the sequences of a set
            cand_g = torch.zeros((int(rows.shape[0]), G), dtype=torch.int32, device=dev).index_add_(
                1, self._const(gidx), bits.to(torch.int32)) > 0
        hitg = torch.nonzero(cand_g.t())                              # [C, 2] (group, row), group-major
        scan_off = off
        if packed_reads is not None and packed_reads.arena is None:
            # the surviving windows as bytes, back to back (each padded to 16 bytes + slack), ascending in the plane
            urows = torch.sort(rows).values
            ulen = length[urows].contiguous()
            ustride = (ulen.to(torch.int64) + (8 + 15)) // 16 * 16
            uends = torch.zeros(int(urows.shape[0]) + 1, dtype=torch.int64, device=dev)
            uends[1:] = torch.cumsum(ustride, 0)
            both = torch.cat([torch.bincount(hitg[:, 0], minlength=G), uends[-1:]]).cpu().numpy()   # the one synchronisation of this stage
            counts, utotal = both[:G], int(both[G])
            arena = torch.empty(utotal + 64, dtype=torch.uint8, device=dev)
            arena[utotal:] = ord("N")
            if urows.numel():
                self.aligner.unpack_windows(packed_reads.plane, packed_reads.exc, off[urows].contiguous(), ulen, arena, uends)
            scan_off = torch.full((n,), -1, dtype=torch.int64, device=dev)
            scan_off[urows] = uends[:-1]
            self.stats["bases_unpacked_after_prefilter"] = self.stats.get("bases_unpacked_after_prefilter", 0) + utotal
        else:
            counts = torch.bincount(hitg[:, 0], minlength=G).cpu().numpy()     # the one synchronisation of this stage
        cjobs, cmeta, pos = [], [], 0
        for g in range(G):
            if counts[g]:
                sel = rows[hitg[pos:pos + int(counts[g]), 1]]
                pos += int(counts[g])
                so, sl = scan_off[sel], length[sel]
                for b in members[g]:
                    cjobs.append((aidx[a_list[b]], so, sl, hint[a_list[b]])); cmeta.append((b, sel))
        if not cjobs:
            return empty
        outs = self._scan_jobs(arena, cjobs, MODE_TWO_PASS, max_len, sort_lengths=ragged, typ_len=typ_len)
        self.stats["pairs_middle_scanned_after_prefilter"] = self.stats.get("pairs_middle_scanned_after_prefilter", 0) + \
            sum(int(j[1].shape[0]) for j in cjobs)
        sb = torch.cat([torch.full((int(sel.shape[0]),), b, dtype=torch.int64, device=dev) for b, sel in cmeta])
        return sb, torch.cat([sel for _, sel in cmeta]), torch.cat(outs)

    def phase_c(self, reads: DeviceReads, start_trim, end_trim, matching: List[int], prove: bool = False,
                prefilter: bool = False) -> MiddleHits:
        """nanopore_read.py:210-243 for every read: adapters in order, and for each adapter keep
        re-aligning against the progressively masked read while the hit reaches --middle_threshold.

        Round 0 aligns every adapter against every (unmasked) trimmed read in one go.  That is
        exact for a read up to and including its first hit; only reads with a hit ("dirty", ~1 %)
        get a private masked copy and continue.  Every later round does the same thing for the
        dirty reads still active: ALL adapters against the current masked state, results consumed
        in the reference's order (from the adapter that just hit, which the reference re-aligns)
        up to and including the next hit.  The alignments consumed are exactly the ones the
        reference's nested loops perform, on the same masked sequences; the ones after a hit are
        speculative and discarded.  Rounds = hits of the most-hit read + 1, not alignments.

        prove=True: round 0 runs the score-only pass for every (read, adapter) pair and the
        traceback only for pairs whose score can still mean an identity >= --middle_threshold
        (identity_score_bound) -- everything else is PROVEN not to be a hit, which is all the
        reference does with those alignments.  Hits, masks and splits are identical; the records of
        the proven non-hits are simply not produced (an option: the default computes them all).

        prefilter=True: the same, with the proof coming from the exact bit-parallel prefilter instead of the
        score pass (pc_prefilter_device: Myers' bit-vector edit distance, one lane per read chunk and adapter; the
        alternative the reference's README.md:355-357 names).  A hit needs full-adapter identity >=
        --middle_threshold, hence at most max_edits(m, threshold) unit-cost edits between the adapter and some
        substring of the read; pairs farther apart are PROVEN not to be hits and never reach the DP.  The reads
        that survive for one of a set's sequences run the two-pass scan for that set (both of its sequences in one
        pass, on the set's ahead-of-time kernel).  Hits, masks, rounds and alignment counts are identical."""
        p = self.p
        dev = self.device
        prove = prove and self.packed_kernels()                # the score bound is derived for the packed kernels' schemes
        ads_sets = self._middle_adapters_with_sets(matching)
        ads = [a for a, _ in ads_sets]
        hint = [("set", si) for _, si in ads_sets]
        self.middle_adapters = ads
        A = len(ads)
        empty = MiddleHits(*(torch.empty(0, dtype=dt, device=dev) for dt in
                             (torch.int64, torch.int32, torch.int32, torch.int32, torch.float64)))
        R = reads.n
        if A == 0 or R == 0:
            return empty
        # masked_seq = seq[start_trim : len - end_trim] with Python slice semantics (nanopore_read.py:56-62)
        # (one round trip: how many reads are left after trimming, and the extremes / mean of their lengths)
        al = self.aligner
        fused = hasattr(al, "round_consume") and os.environ.get("PC_NO_FUSED_GLUE", "0") in ("", "0")     # pc_middle.hip: one launch each
        if fused:
            toff, tlen, st4 = al.trim_windows(reads.off, reads.length, start_trim, end_trim)
            mm = st4.cpu()
            mm[2] = (al.STAT_BIG - mm[2]) if int(mm[2]) else 0
            pos_len = None
        else:
            s_pos, e_pos = trimmed_interval(reads.length, start_trim, end_trim)
            tlen = torch.clamp(e_pos - s_pos, min=0).to(torch.int32)
            toff = reads.off + s_pos
            pos_len = tlen > 0
            big = torch.iinfo(torch.int32).max
            mm = torch.stack([pos_len.sum(), tlen.max(), torch.where(pos_len, tlen, torch.full_like(tlen, big)).min(),
                              tlen.sum(dtype=torch.int64)]).cpu()
        n_live = int(mm[0])
        if n_live == 0:
            return empty
        if n_live == R:
            live = torch.arange(R, device=dev)
            loff, llen = toff, tlen
        else:
            if pos_len is None:
                pos_len = tlen > 0
            live = torch.nonzero_static(pos_len, size=n_live).flatten() if hasattr(torch, "nonzero_static") else torch.nonzero(pos_len).flatten()
            loff, llen = toff[live], tlen[live]
        packed_only = reads.arena is None
        if packed_only and not prefilter:
            reads.materialize(self.aligner)                          # every pair runs the DP: every base is needed as a byte
            packed_only = False
        # (windows of different lengths are handed over longest first even when they differ by a per cent -- 8-kb reads that lost
        # 0..150 bases to their trims: tiles of nearly one length keep the specialised score kernel on its block-resolved path;
        # skipping the sort for such batches was measured: 0.3 ms of glue saved, 1.6 ms of scan lost at 1 M reads)
        max_len, ragged, typ_len = int(mm[1]), bool(int(mm[1]) != int(mm[2])), int(mm[3]) // n_live
        aidx = [self.seq_index[a[1]] for a in ads]

        def identity_of(rec):
            full, _ = _identities(rec)
            return torch.where(rec[..., 0] == -1, torch.zeros_like(full), full)

        # ---- round 0: all adapters x all reads, unmasked -------------------------------------
        jobs0 = [(ai, loff, llen, h) for ai, h in zip(aidx, hint)]
        bounds = [self.identity_score_bound(len(self.seqs[ai]), p.middle_threshold) for ai in aidx] if prove else None
        sparse0 = None
        if prefilter:
            ks = [self.aligner.max_edits(len(self.seqs[ai]), p.middle_threshold) for ai in aidx]
            sparse0 = self._prefiltered_scan(reads.arena, loff, llen, max_len, list(range(A)), aidx, hint, ks, ragged, typ_len,
                                             packed_reads=reads if packed_only else None)
            packed_only = packed_only and reads.arena is None         # (the packed route may have been refused: everything unpacked)
        elif prove and all(b is not None for b in bounds):
            score = torch.stack(self._scan_jobs(reads.arena, jobs0, MODE_SCORE, max_len, sort_lengths=ragged, typ_len=typ_len))[:, :, 4]      # [A, L]
            cand = torch.nonzero(score >= self._const(bounds)[:, None])                     # adapter-major
            counts = torch.bincount(cand[:, 0], minlength=A).cpu().numpy()
            L = int(live.numel())
            recs = torch.zeros((A, L, RESULT_INTS), dtype=torch.int32, device=dev)
            cjobs, csel, pos = [], [], 0
            for a in range(A):
                if counts[a]:
                    sel = cand[pos:pos + int(counts[a]), 1]
                    pos += int(counts[a])
                    cjobs.append((aidx[a], loff[sel], llen[sel])); csel.append((a, sel))     # (different windows per adapter: never fused)
            if cjobs:
                for (a, sel), o in zip(csel, self._scan_jobs(reads.arena, cjobs, MODE_TWO_PASS, max_len, sort_lengths=ragged, typ_len=typ_len)):
                    recs[a, sel] = o
            outs = [recs[a] for a in range(A)]
            # an all-zero record (rs = 0, lengths 0) is "not a hit" below: 0/0 identities are masked
            fulls = torch.stack([torch.nan_to_num(identity_of(rec), nan=0.0) for rec in outs])
            self.stats["pairs_middle_traced_after_proof"] = self.stats.get("pairs_middle_traced_after_proof", 0) + int(counts.sum())
        else:
            outs = self._scan_jobs(reads.arena, jobs0, MODE_TWO_PASS, max_len, sort_lengths=ragged, typ_len=typ_len)
            if fused:
                fulls, hit0_f = al.middle_hits(torch.stack(outs), p.middle_threshold)          # [A, L] each
            else:
                fulls = torch.stack([identity_of(rec) for rec in outs])  # [A, L]
        L_ = int(live.numel())
        if sparse0 is not None:
            sa, sw, sr = sparse0                                     # every pair not listed is proven not to be a hit
            if fused:
                full_s, hit_s = al.middle_hits(sr.contiguous(), p.middle_threshold)
            else:
                full_s = torch.nan_to_num(identity_of(sr), nan=0.0)
                hit_s = (full_s >= p.middle_threshold) & (sr[:, 0] != -1)
            dmask = torch.zeros(L_, dtype=torch.int32, device=dev).index_add_(0, sw, hit_s.to(torch.int32)) > 0
        elif fused and not prove:
            dmask = hit0_f.any(dim=0)
        else:
            hit0 = (fulls >= p.middle_threshold) & torch.stack([rec[:, 0] != -1 for rec in outs])
            dmask = hit0.any(dim=0)
        # (one round trip: the dirty reads -- those with a hit -- their number, longest, total padded size and mean length)
        dstride_all = (llen.to(torch.int64) + (8 + 15)) // 16 * 16
        zero64 = torch.zeros((), dtype=torch.int64, device=dev)
        mm2 = torch.stack([dmask.sum(), torch.where(dmask, llen.to(torch.int64), zero64).max(), torch.where(dmask, dstride_all, zero64).sum(),
                           torch.where(dmask, llen.to(torch.int64), zero64).sum()]).cpu()
        Dn = int(mm2[0])
        if Dn > 0:                                                   # dirty reads (indices into live), increasing
            d_sel = torch.nonzero_static(dmask, size=Dn).flatten() if hasattr(torch, "nonzero_static") else torch.nonzero(dmask).flatten()
        n_align = A * int(live.numel())                              # alignments the reference performs
        n_spec = 0                                                   # speculative ones, discarded
        rounds = 0
        H_read, H_ad, H_s, H_e, H_id = [], [], [], [], []
        if Dn > 0:
            if sparse0 is not None:                                  # [A, Dn, 8] from the sparse records of the dirty reads
                # (no boolean masks -- each is a nonzero, a host round trip: the records of reads that are not dirty all go to
                # one spare row behind the table)
                where = torch.full((int(live.numel()),), -1, dtype=torch.int64, device=dev)
                where[d_sel] = torch.arange(Dn, device=dev)
                w2 = where[sw]
                flat = torch.where(w2 >= 0, sa * Dn + w2, torch.full_like(w2, A * Dn))
                rec_flat = torch.zeros((A * Dn + 1, RESULT_INTS), dtype=torch.int32, device=dev)
                full_flat = torch.zeros(A * Dn + 1, dtype=torch.float64, device=dev)
                rec_flat[flat] = sr
                full_flat[flat] = full_s
                rec_all = rec_flat[:A * Dn].view(A, Dn, RESULT_INTS)
                full_all = full_flat[:A * Dn].view(A, Dn)
            else:
                rec_all = torch.stack([o[d_sel] for o in outs])      # [A, Dn, 8] for the current masked state
                full_all = fulls[:, d_sel]
            # private, maskable copies of the dirty reads, back to back whatever their lengths (each padded
            # with N to a multiple of 16 bytes plus slack: the kernels read whole dwords)
            dlen = llen[d_sel].contiguous()
            dstride = (dlen.to(torch.int64) + (8 + 15)) // 16 * 16
            d_ends = torch.zeros(Dn + 1, dtype=torch.int64, device=dev)
            d_ends[1:] = torch.cumsum(dstride, 0)
            dmax, dtotal, dtyp = int(mm2[1]), int(mm2[2]), int(mm2[3]) // Dn
            dirty = torch.empty(dtotal + 64, dtype=torch.uint8, device=dev)
            dirty[dtotal:] = ord("N")
            d_off = d_ends[:-1]
            if packed_only:                                         # (d_sel is ascending: torch.unique)
                self.aligner.unpack_windows(reads.plane, reads.exc, loff[d_sel].contiguous(), dlen, dirty, d_ends, ord("N"))
            elif hasattr(self.aligner, "copy_windows"):
                self.aligner.copy_windows(reads.arena, loff[d_sel].contiguous(), dlen, dirty, d_ends, ord("N"))
            else:                                                    # injected test aligner: the same copy in torch
                seg = torch.repeat_interleave(torch.arange(Dn, device=dev), dstride)
                pos = torch.arange(dtotal, device=dev, dtype=torch.int64) - d_off[seg]
                inside = pos < dlen[seg]
                src = loff[d_sel][seg] + torch.clamp(pos, max=(dlen.to(torch.int64) - 1)[seg])
                dirty[:dtotal] = torch.where(inside, reads.arena[src], torch.full_like(src, ord("N"), dtype=torch.uint8))
            arow = torch.arange(A, device=dev)[:, None]
            cur = torch.zeros(Dn, dtype=torch.int64, device=dev)     # adapter each dirty read is at
            act = torch.arange(Dn, device=dev)
            n_align -= A * Dn                                        # re-counted below as consumed
            scheduled = A * Dn
            while True:
                # consume, per active read: adapters cur.. in order, up to and including the first hit
                # (one round trip per round: alignments consumed, reads that hit, the first adapter among them, bases to mask)
                if fused:
                    anyh, a_hit, cnt_all, st4 = al.round_consume(full_all, rec_all, cur, act, p.middle_threshold)
                    st_ = st4.cpu()
                    n_used, n_hit, n_mask = int(st_[0]), int(st_[1]), int(st_[3])
                    a0 = (al.STAT_BIG - int(st_[2])) if int(st_[2]) else A
                    r_all = None
                else:
                    c = cur[act]
                    hm = (full_all[:, act] >= p.middle_threshold) & (rec_all[:, act, 0] != -1) & (arow >= c[None, :])
                    anyh = hm.any(dim=0)
                    a_hit = hm.to(torch.int32).argmax(dim=0)
                    used = torch.where(anyh, a_hit - c + 1, A - c)
                    r_all = rec_all[a_hit, act]
                    cnt_all = torch.where(anyh, torch.clamp(r_all[:, 1] + 1 - r_all[:, 0], min=0), torch.zeros_like(r_all[:, 0])).to(torch.int64)
                    st_ = torch.stack([used.sum(), anyh.sum(), torch.where(anyh, a_hit, torch.full_like(a_hit, A)).min().to(torch.int64),
                                       cnt_all.sum()]).cpu()
                    n_used, n_hit, a0, n_mask = int(st_[0]), int(st_[1]), int(st_[2]), int(st_[3])
                n_align += n_used
                n_spec += scheduled - n_used
                if n_hit == 0:
                    break
                hidx = torch.nonzero_static(anyh, size=n_hit).flatten() if hasattr(torch, "nonzero_static") else torch.nonzero(anyh).flatten()
                hsel = act[hidx]
                ah = a_hit[hidx].to(torch.int64)
                r = rec_all[ah, hsel] if r_all is None else r_all[hidx]
                rs, re = r[:, 0], r[:, 1] + 1
                H_read.append(live[d_sel[hsel]]); H_ad.append(ah.to(torch.int32))
                H_s.append(rs); H_e.append(re); H_id.append(full_all[ah, hsel])
                # masked_seq[rs:re] = '-' * n: the masked positions of all hits as one index list
                cnt = cnt_all[hidx]
                first = torch.cumsum(cnt, 0) - cnt
                run = torch.repeat_interleave(d_off[hsel] + rs.to(torch.int64) - first, cnt, output_size=n_mask)
                dirty.index_fill_(0, run + torch.arange(n_mask, device=dev, dtype=torch.int64), ord("-"))   # (x[idx] = scalar uploads the scalar: a round trip)
                cur[hsel] = ah                                       # the reference re-aligns the adapter that hit
                act = hsel
                rounds += 1
                o_act, l_act = d_off[act], dlen[act]
                if prefilter:                                        # the masked reads go through the same proof first
                    rb, rw, rr = self._prefiltered_scan(dirty, o_act, l_act, dmax, list(range(a0, A)), aidx, hint, ks, ragged, dtyp)
                    outs_r = torch.zeros((A - a0, int(act.numel()), RESULT_INTS), dtype=torch.int32, device=dev)
                    outs_r[rb, rw] = rr
                else:
                    outs_r = self._scan_jobs(dirty, [(aidx[a], o_act, l_act, hint[a]) for a in range(a0, A)], MODE_TWO_PASS, dmax,
                                             sort_lengths=ragged, typ_len=dtyp)
                scheduled = (A - a0) * int(act.numel())
                if not torch.is_tensor(outs_r):
                    outs_r = torch.stack(outs_r)                     # [A - a0, active, 8]
                # (all adapters at once: a loop over the 196 sequences of a barcode panel is ~2 000 tiny launches a round)
                rec_all[a0:, act] = outs_r
                full_all[a0:, act] = al.middle_hits(outs_r.contiguous(), p.middle_threshold)[0] if fused else torch.nan_to_num(identity_of(outs_r), nan=0.0)
        self.stats["pairs_middle"] += n_align
        self.stats["pairs_middle_speculative"] = self.stats.get("pairs_middle_speculative", 0) + n_spec
        if not H_read:
            empty.rounds, empty.alignments = rounds, n_align
            return empty
        return MiddleHits(torch.cat(H_read), torch.cat(H_ad), torch.cat(H_s), torch.cat(H_e), torch.cat(H_id),
                          rounds, n_align)

    def close(self):
        self.aligner.close()
