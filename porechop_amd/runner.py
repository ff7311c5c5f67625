"""End to end: reads in -> adapters found, ends trimmed, barcodes called, chimeras split -> reads out.

This is the batch form of everything porechop/porechop.py:33-79 (main) does around the hot path --
SURVEY.md section 8f rows 2 (per-read orchestration) and 3 (output) -- written against arrays
instead of per-read Python objects:

  loading            porechop.py:216-268   (file, or Albacore directory with per-file check reads
                                            and barcode calls read off the path)  -> io.ReadSet (C++)
  phase A + rules    porechop.py:286-327, 374-390, 330-371, 410-436              -> Pipeline.phase_a, panel.py
  phase B            porechop.py:438-514 + nanopore_read.py:166-208              -> Pipeline.phase_b
  barcode calls      nanopore_read.py:399-473 (determine_barcode)                -> call_barcodes (tensor ops)
  phase C            porechop.py:533-595 + nanopore_read.py:210-243              -> Pipeline.phase_c
  pieces to write    nanopore_read.py:56-147 (trim slices, split parts, naming)  -> plan_output (numpy)
  writing            porechop.py:607-734                                         -> ReadSet.write (C++)

The alignments come from the GPU library only (Pipeline's default aligner); output files are
byte-identical to the reference's (tests/test_runner_*.py).  Progress tables and coloured
per-read dumps (verbosity >= 1 in the reference) are not reproduced: run() returns the numbers.
"""
import os
import re
import shutil
import tempfile
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import panel as panel_rules
from .distributed import gather_in_order, reduce_presence, shard_by_bases
from .io import GzStream, ReadSet, gz_finish
from .pipeline import AdapterSet, DeviceReads, Pipeline, ScanParams, trimmed_interval
from .pipeline import call_barcodes as _call_barcodes


@dataclass
class Options:
    """Defaults of porechop/porechop.py:85-185 (get_arguments)."""
    format: str = "auto"                      # auto | fasta | fastq | fasta.gz | fastq.gz
    barcode_threshold: float = 75.0
    barcode_diff: float = 5.0
    require_two_barcodes: bool = False
    untrimmed: bool = False
    discard_unassigned: bool = False
    adapter_threshold: float = 90.0
    check_reads: int = 10000
    scoring_scheme: Tuple[int, int, int, int] = (3, -6, -5, -2)
    end_size: int = 150
    min_trim_size: int = 4
    extra_end_trim: int = 2
    end_threshold: float = 75.0
    no_split: bool = False
    discard_middle: bool = False
    middle_threshold: float = 90.0
    extra_middle_trim_good_side: int = 10
    extra_middle_trim_bad_side: int = 100
    min_split_read_size: int = 1000


@dataclass
class RunResult:
    n_reads: int = 0
    read_type: str = "FASTQ"
    matching_sets: List[str] = field(default_factory=list)       # after the 1D^2 fix-up and the full barcode sets
    barcode_orientation: Optional[str] = None
    start_trim: Optional[np.ndarray] = None                      # int32 [R]
    end_trim: Optional[np.ndarray] = None
    barcode_calls: Optional[List[str]] = None                    # per read, when binning
    middle_hit_reads: int = 0
    files: Dict[str, Tuple[int, int]] = field(default_factory=dict)   # path -> (records written, bases)
    out_format: str = "fastq"
    seconds: Dict[str, float] = field(default_factory=dict)      # wall-clock per stage
    counts: Dict[str, int] = field(default_factory=dict)         # whole-run tallies where start_trim / end_trim hold one rank's reads
    # a sharded run (run_sharded): start_trim / end_trim / barcode_calls cover THIS rank's reads only -- reads
    # [first_read, first_read + local_reads) of the n_reads of the file; everywhere else first_read = 0, local_reads = n_reads
    first_read: int = 0
    local_reads: Optional[int] = None


# (read, adapter) pairs one block of phases B / C may hold at a time: 8 ints each, a few copies -> a few GB of HBM
READ_BLOCK_PAIRS = 64_000_000
MIN_READ_BLOCK = 4096


class UsageError(ValueError):
    """What the reference reports with sys.exit('Error: ...')."""


def _albacore_barcode(path):
    # porechop.py:271-278
    if "/unclassified/" in path:
        return "none"
    m = re.findall(r"/barcode(\d\d)/", path)
    if m:
        return "BC" + m[-1]
    return None


def _load(input_path, check_read_count):
    """-> (ReadSet, check-read indices, per-read Albacore call or None)."""
    if os.path.isfile(input_path):
        rs = ReadSet(input_path)
        return rs, np.arange(min(rs.count, max(0, check_read_count)), dtype=np.int64), None
    if os.path.isdir(input_path):
        fastqs = sorted(os.path.join(d, f) for d, _, fs in os.walk(input_path) for f in fs
                        if f.lower().endswith(".fastq") or f.lower().endswith(".fastq.gz"))
        if not fastqs:
            raise UsageError("Error: could not find fastq files in " + input_path)
        rs = ReadSet(fastqs)
        if not rs.is_fastq:
            raise UsageError("Error: " + fastqs[0] + " is not FASTQ")
        per_file = int(round(check_read_count / len(fastqs)))
        fi = rs.file_index
        first_of_file = np.searchsorted(fi, np.arange(len(fastqs)), side="left")
        rank_in_file = np.arange(rs.count, dtype=np.int64) - first_of_file[fi]
        check = np.nonzero(rank_in_file < per_file)[0].astype(np.int64)
        calls = [_albacore_barcode(p) for p in fastqs]
        return rs, check, [calls[i] for i in fi]
    raise UsageError("Error: could not find " + input_path)


def call_barcodes(names: List[str], start_scores: torch.Tensor, end_scores: torch.Tensor, opts: Options) -> np.ndarray:
    """nanopore_read.py:399-466 for every read at once: see pipeline.call_barcodes (names[k] is bin k's name)."""
    return _call_barcodes(len(names), start_scores, end_scores, opts.barcode_threshold, opts.barcode_diff,
                          opts.require_two_barcodes)


def barcode_bins(pl, bc_sets):
    """-> (bin names, [(start set, end set)] per bin).  The reference's two score dicts are keyed by bin
    name: a later set with the same name overwrites the value but keeps the first insertion's position."""
    names = _bin_names(pl, bc_sets)
    col = {n: k for k, n in enumerate(names)}
    bins = [[None, None] for _ in names]
    for i in bc_sets:
        k = col[panel_rules.barcode_name(pl.sets[i])]
        if pl.sets[i].start is not None:
            bins[k][0] = i
        if pl.sets[i].end is not None:
            bins[k][1] = i
    return names, [tuple(b) for b in bins]


def _bin_names(pl, bc_sets):
    names = []
    for i in bc_sets:
        n = panel_rules.barcode_name(pl.sets[i])
        if n not in names:
            names.append(n)
    return names


def _barcode_bin_names(pl, match_idx, orientation):
    return _bin_names(pl, [i for i in match_idx if panel_rules.is_barcode(pl.sets[i])
                           and panel_rules.barcode_direction(pl.sets[i]) == orientation])


def _split_parts(tlen, intervals, min_size):
    """get_split_read_parts (nanopore_read.py:76-95): positions of the trimmed read not covered by
    any [trim_start, trim_end) -> maximal runs -> those of at least min_size bases."""
    cover = np.zeros(tlen + 1, dtype=np.int32)
    for a, b in intervals:
        a, b = max(a, 0), min(b, tlen)
        if b > a:
            cover[a] += 1
            cover[b] -= 1
    keep = np.cumsum(cover[:tlen]) == 0
    edges = np.diff(np.concatenate([[0], keep.astype(np.int8), [0]]))
    starts, ends = np.nonzero(edges == 1)[0], np.nonzero(edges == -1)[0]
    return [(int(s), int(e - s)) for s, e in zip(starts, ends) if e - s >= min_size]


def _resolve_format(opts: Options, output, barcode_dir, read_type, input_path):
    # porechop.py:627-655
    fmt = opts.format
    if fmt == "auto":
        if output is None:
            fmt = read_type.lower()
            if barcode_dir is not None and input_path.lower().endswith(".gz"):
                fmt += ".gz"
        elif ".fasta.gz" in output.lower():
            fmt = "fasta.gz"
        elif ".fastq.gz" in output.lower():
            fmt = "fastq.gz"
        elif ".fasta" in output.lower():
            fmt = "fasta"
        elif ".fastq" in output.lower():
            fmt = "fastq"
        else:
            fmt = read_type.lower()
    gz = fmt.endswith(".gz") and (barcode_dir is not None or output is not None)
    if gz:
        fmt = fmt[:-3]
    # (to stdout a '.gz' format is left as it is and, not being 'fasta', prints FASTQ -- as there)
    return fmt, gz


def _is_gzip(path):
    # misc.py:60-81 get_compression_type: by magic bytes, not by name
    try:
        with open(path, "rb") as f:
            return f.read(3) == b"\x1f\x8b\x08"
    except OSError:
        return False


def gz_out_hint(opts, output, barcode_dir, input_path):
    """Will this run write gzip?  (_resolve_format before the read type is known: FASTQ assumed.)"""
    try:
        return _resolve_format(opts, output, barcode_dir, "FASTQ", input_path)[1]
    except Exception:
        return False


def _emit(rs, pr, ps_, pn_, num, pf, paths, fastq, file_pos, gz, shared=False):
    """Pieces of one read set into their files at file_pos (updated): plain bytes (pc_readset_write_at / _shared), or --
    gz -- formatted and deflated by all cores in memory, then written (pc_readset_compress: what the reference gets from
    `pigz -p <threads>` over a temporary file, porechop.py:640-651,685-729).  A gz file is ended by io.gz_finish."""
    if not gz:
        (rs.write_shared if shared else rs.write_at)(pr, ps_, pn_, num, pf, paths, fastq, file_pos)
        return
    img = rs.compress(pr, ps_, pn_, num, pf, len(paths), fastq)
    try:
        img.write(paths, file_pos, shared=shared)
    finally:
        img.close()


def _find_sets(pl, panel, reads, check_idx, opts, barcode_dir, sharded=False):
    """Phase A over the check reads + the set-level rules (porechop.py:286-436) -> (matching sets incl. the full
    barcode sets, their indices in pl.sets, barcode orientation or None)."""
    dev = pl.device
    if reads is not None and check_idx.size:
        # the pruned search gives the same matching sets and best identities, but the table
        # entries of the side of a set that stays below the threshold are only lower bounds; the
        # barcode-kit choice (porechop.py:343-366) sums BOTH sides of every matching barcode set
        # in its tie-break, so a binning run needs the exact table
        bs, be = pl.phase_a(reads, torch.from_numpy(np.ascontiguousarray(check_idx)).to(dev), prune=barcode_dir is None)
    else:
        bs = torch.zeros(len(panel), dtype=torch.float64, device=dev)
        be = torch.zeros(len(panel), dtype=torch.float64, device=dev)
    if sharded:
        bs, be = reduce_presence(bs, be)                    # nanopore_read.py:159,164 across all ranks' check reads
    bs, be = bs.cpu().numpy(), be.cpu().numpy()
    index_of = {id(s): i for i, s in enumerate(pl.sets)}
    score = lambda s: max(bs[index_of[id(s)]], be[index_of[id(s)]])
    matching = [s for s in panel if "(full sequence)" not in s.name and score(s) >= opts.adapter_threshold]
    matching = panel_rules.fix_up_1d2(matching, score)
    orientation = None
    if barcode_dir is not None:
        try:
            orientation = panel_rules.choose_barcoding_kit(matching, lambda s: bs[index_of[id(s)]],
                                                           lambda s: be[index_of[id(s)]])
        except panel_rules.NoBarcodes as e:
            raise UsageError(str(e))
    with_full = panel_rules.add_full_barcode_sets(panel, matching)
    pl.add_sets(with_full[len(matching):])
    index_of = {id(s): i for i, s in enumerate(pl.sets)}
    matching = with_full
    return matching, [index_of[id(s)] for s in matching], orientation


def _scan_reads(pl, reads, R, match_idx, opts, barcode_dir, orientation, lap=lambda *a, **k: None):
    """Phases B (+ barcode calls) and C for R resident reads -> (start_trim, end_trim [device int32], bin index per
    read [numpy int64, -1 = none], MiddleHits or None, bin names)."""
    dev = pl.device
    start_trim = torch.zeros(R, dtype=torch.int32, device=dev)
    end_trim = torch.zeros(R, dtype=torch.int32, device=dev)
    hits = None
    ci = np.full(R, -1, dtype=np.int64)
    names = []
    if match_idx and R:
        # ---- phase B (+ barcode calls) ------------------------------------------------
        check_barcodes = barcode_dir is not None
        bc_sets = [i for i in match_idx if check_barcodes and panel_rules.is_barcode(pl.sets[i])
                   and panel_rules.barcode_direction(pl.sets[i]) == orientation]
        # Phases B and C are per read: they run over blocks of reads so that the scratch of a run
        # (8 ints per (read, adapter) pair, all middle adapters at once) is bounded by the block, not by
        # the input -- the reference holds one read's alignments at a time (nanopore_read.py:149-243).
        if check_barcodes:
            names, bins = barcode_bins(pl, bc_sets)
        n_mid = max(1, len(pl.middle_adapter_list(match_idx)))
        n_end = max(1, 2 * len(match_idx))
        block = max(MIN_READ_BLOCK, int(READ_BLOCK_PAIRS // max(n_mid, n_end)))
        st_parts, et_parts, ci_parts, hit_parts = [], [], [], []
        for b0 in range(0, R, block):
            b1 = min(R, b0 + block)
            sub = reads if (b0 == 0 and b1 == R) else DeviceReads(reads.arena, reads.off[b0:b1], reads.length[b0:b1])
            if check_barcodes:
                st_b, et_b, ci_b = pl.phase_b_demux(sub, match_idx, bins, opts.barcode_threshold, opts.barcode_diff,
                                                    opts.require_two_barcodes)
                ci_parts.append(ci_b)
            else:
                st_b, et_b = pl.phase_b(sub, match_idx)
            lap("phase_b", sync=True)
            if not opts.no_split:
                # identical hits either way: behind the exact prefilter on the GPU library (most pairs never reach the
                # DP), behind the score bound with an injected test aligner
                fast = getattr(pl.aligner, "fast_prefilter", False)
                hb = pl.phase_c(sub, st_b, et_b, match_idx, prove=not fast, prefilter=fast)
                if hb.read.numel():
                    hb.read = hb.read + b0
                    hit_parts.append(hb)
                lap("phase_c", sync=True)
            st_parts.append(st_b); et_parts.append(et_b)
        start_trim, end_trim = torch.cat(st_parts), torch.cat(et_parts)
        if check_barcodes:
            ci = np.concatenate(ci_parts)
        if not opts.no_split:
            from .pipeline import MiddleHits
            if hit_parts:
                hits = MiddleHits(*(torch.cat([getattr(h_, f) for h_ in hit_parts]) for f in ("read", "adapter", "start", "end", "identity")),
                                  max(h_.rounds for h_ in hit_parts), sum(h_.alignments for h_ in hit_parts))
            else:
                hits = MiddleHits(*(torch.empty(0, dtype=dt, device=dev) for dt in
                                    (torch.int64, torch.int32, torch.int32, torch.int32, torch.float64)))
    return start_trim, end_trim, ci, hits, names


def _plan_pieces(opts, lengths, st, et, h, calls, barcode_dir, discard_middle, matching, pl, match_idx):
    """Which pieces of which reads are written (porechop.py:607-734, nanopore_read.py:76-147): lengths / st / et per
    read (numpy), h = middle hits [H, 4] (read, adapter, start, end in trimmed-read coordinates), calls = bin name
    per read or None -> (piece_read, piece_start, piece_len, piece_number, trimmed lengths, reads with middle hits)."""
    s_pos, e_pos = trimmed_interval(torch.from_numpy(np.ascontiguousarray(lengths).copy()), torch.from_numpy(st), torch.from_numpy(et))
    s_pos, e_pos = s_pos.numpy(), e_pos.numpy()
    tlen = np.maximum(e_pos - s_pos, 0)
    split_of = {}
    if h.shape[0]:
        start_names = {s.start[0] for s in matching if s.start is not None}
        end_names = {s.end[0] for s in matching if s.end is not None}
        good, bad = opts.extra_middle_trim_good_side, opts.extra_middle_trim_bad_side
        ad_names = [a[0] for a in pl.middle_adapter_list(match_idx)]
        lo = np.array([bad if n in start_names else good for n in ad_names], dtype=np.int64)
        hi = np.array([bad if n in end_names else good for n in ad_names], dtype=np.int64)
        for r, a, s, e in h:
            split_of.setdefault(int(r), []).append((int(s - lo[a]), int(e + hi[a])))
    whole = opts.untrimmed
    p_start = np.where(whole, 0, s_pos).astype(np.int64)
    p_len = np.where(whole, lengths, tlen).astype(np.int64)
    emit = p_len > 0                                            # "Don't return empty sequences"
    if split_of:
        emit[np.fromiter(split_of.keys(), dtype=np.int64)] = False
    if barcode_dir is not None and opts.discard_unassigned:
        emit &= np.array([c != "none" for c in calls], dtype=bool)
    # reads with middle hits: dropped when discarding, otherwise split and numbered
    extra = []                                                  # (read, start, len, number)
    if split_of and not discard_middle:
        for r, ivs in split_of.items():
            if barcode_dir is not None and opts.discard_unassigned and calls[r] == "none":
                continue
            for k, (ps, pn) in enumerate(_split_parts(int(tlen[r]), ivs, opts.min_split_read_size)):
                extra.append((r, int(s_pos[r]) + ps, pn, k + 1))
    base_reads = np.nonzero(emit)[0].astype(np.int64)
    if extra:
        ex = np.array(extra, dtype=np.int64)
        pr = np.concatenate([base_reads, ex[:, 0]])
        ps_ = np.concatenate([p_start[base_reads], ex[:, 1]])
        pn_ = np.concatenate([p_len[base_reads], ex[:, 2]])
        num = np.concatenate([np.zeros(base_reads.size, dtype=np.int64), ex[:, 3]])
        order = np.lexsort((num, pr))                           # read order, pieces of a read in order
        pr, ps_, pn_, num = pr[order], ps_[order], pn_[order], num[order]
    else:
        pr, ps_, pn_, num = base_reads, p_start[base_reads], p_len[base_reads], np.zeros(base_reads.size, dtype=np.int64)
    return pr, ps_, pn_, num, tlen, len(split_of)


# A plain FASTQ file larger than this is run as a stream of blocks (run_streamed); PC_STREAM_BLOCK_BYTES overrides
STREAM_BLOCK_BYTES = 1 << 28


def _stream_block_bytes():
    v = os.environ.get("PC_STREAM_BLOCK_BYTES")
    return int(v) if v else STREAM_BLOCK_BYTES


def run_streamed(input_path, output, barcode_dir, opts: Options, device=None, aligner=None,
                 adapter_panel: List[AdapterSet] = None, block_bytes: int = None) -> Optional[RunResult]:
    """run() for a plain FASTQ file as a STREAM of blocks: a loader thread parses block k+1 (pc_readset_load_segment,
    all host cores) while block k is uploaded and scanned on the GPU and a writer thread formats and writes block k-1
    (pc_readset_write_at) -- host memory holds three blocks instead of the input, and ingest, scan and writing overlap.
    Phase A and the set-level rules run once, on the first block, which must hold the check reads.  Everything after
    that is per read in Porechop (phases B and C, barcode calls, splitting, naming), so the output files are the ones
    run() writes.  Plain FASTA files (cut where a line begins with '>') and the gzip forms of both stream the same way.
    -> None when the input is not streamable (a directory, irregular records, a damaged gzip stream, or a first block
    without the check reads): the caller loads the whole file."""
    import queue
    import threading
    block_bytes = block_bytes or _stream_block_bytes()
    size = os.path.getsize(input_path)
    t_start = time.perf_counter()
    # .gz input: a producer thread inflates ahead (sized members on several cores, anything else through zlib) and hands
    # over the same blocks the plain file would be cut into (io.GzStream / pc_gzstream_next)
    gz_in = GzStream(input_path) if _is_gzip(input_path) else None
    # the first block is made large enough to hold the check reads (phase A looks at the first N reads of the file)
    first_bytes = block_bytes
    pos = 0
    if gz_in is not None:
        first = gz_in.next(block_bytes, max(0, opts.check_reads))
        if first is False or first is None:      # neither a regular 4-line FASTQ nor a FASTA stream (damaged, empty): the whole-file loader's case
            gz_in.close()
            return None
    while gz_in is None:
        first, pos = ReadSet.segment(input_path, 0, first_bytes)
        if first is None:
            return None
        if pos >= size or first.count >= max(0, opts.check_reads):
            break
        first.close()
        first_bytes *= 4
    discard_middle = opts.discard_middle or barcode_dir is not None
    res = RunResult(n_reads=0, read_type="FASTQ" if first.is_fastq else "FASTA")     # (plain FASTA streams too: cut at '>' lines)
    busy = {"load": time.perf_counter() - t_start, "scan": 0.0, "write": 0.0}

    panel = list(adapter_panel) if adapter_panel is not None else panel_rules.load_panel()
    params = ScanParams(end_size=opts.end_size, min_trim_size=opts.min_trim_size, extra_end_trim=opts.extra_end_trim,
                        end_threshold=opts.end_threshold, middle_threshold=opts.middle_threshold,
                        adapter_threshold=opts.adapter_threshold, check_reads=opts.check_reads,
                        scores=tuple(int(x) for x in opts.scoring_scheme))
    pl = Pipeline(panel, params, device=device, aligner=aligner)
    dev = pl.device
    if aligner is None:
        pl.aligner.lib.pc_jit_async(1)

    fmt, gz = _resolve_format(opts, output, barcode_dir, res.read_type, input_path)
    res.out_format = fmt
    fastq = fmt != "fasta"
    whole = opts.untrimmed
    if barcode_dir is not None:
        os.makedirs(barcode_dir, exist_ok=True)
        target = None
    elif output is None:
        target = "-"
    else:
        target = output                  # (gz: written compressed as it goes, no temporary file)
    ext = "." + fmt + (".gz" if gz else "")

    # ---- loader: blocks 1.. (block 0 is in hand) ------------------------------------------------------
    loaded = queue.Queue(maxsize=1)
    failure = []
    stop = threading.Event()

    # the loader and the writer work at the same time: each gets about half of the cores this process may use
    from ._lib import load_library
    io_lib = load_library()
    try:
        ncores = len(os.sched_getaffinity(0))
        with open("/sys/fs/cgroup/cpu.max") as f_:
            quota, period = f_.read().split()
        if quota != "max":
            ncores = min(ncores, max(1, int(int(quota) / int(period))))
    except Exception:
        ncores = os.cpu_count() or 2
    share = max(2, min(32, ncores // 2))
    # gz output: deflating is the wall of the run (measured, 16 cores: the writer busy 3.6 s of 3.8 at half the cores, the
    # loader -- even one that inflates -- 1.4 s): three quarters of the cores to the writer
    share_w = max(2, min(48, ncores * 3 // 4)) if gz_out_hint(opts, output, barcode_dir, input_path) else share
    share_l = max(2, ncores - share_w) if share_w != share else share

    def loader():
        p_ = pos
        io_lib.pc_io_set_thread_limit(share_l)
        try:
            while gz_in is not None and not stop.is_set():
                t0 = time.perf_counter()
                rs_ = gz_in.next(block_bytes)
                busy["load"] += time.perf_counter() - t0
                if rs_ is None:
                    break
                if rs_ is False:
                    raise ValueError("Error: " + input_path + " could not be parsed - is it formatted correctly?")
                loaded.put(rs_)
            while gz_in is None and p_ < size and not stop.is_set():
                t0 = time.perf_counter()
                rs_, nxt = ReadSet.segment(input_path, p_, block_bytes)
                busy["load"] += time.perf_counter() - t0
                if rs_ is None or nxt <= p_:
                    raise ValueError("Error: " + input_path + " could not be parsed - is it formatted correctly?")
                p_ = nxt
                loaded.put(rs_)
        except BaseException as e:           # noqa: handed to the main thread
            failure.append(e)
        loaded.put(None)

    # ---- writer ---------------------------------------------------------------------------------------
    to_write = queue.Queue(maxsize=1)
    paths, file_pos, stats = [], np.zeros(0, dtype=np.int64), {}

    def writer():
        nonlocal file_pos
        io_lib.pc_io_set_thread_limit(share_w)
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                rs_, pr, ps_, pn_, num, bins_of_piece, tlen = item
                t0 = time.perf_counter()
                if barcode_dir is not None:
                    pf = np.zeros(pr.size, dtype=np.int32)
                    for b in sorted(set(bins_of_piece)):
                        path = os.path.join(barcode_dir, b + ext)
                        if path not in paths:
                            paths.append(path)
                            file_pos = np.concatenate([file_pos, np.zeros(1, dtype=np.int64)])
                    index = {p_: k for k, p_ in enumerate(paths)}
                    pf = np.fromiter((index[os.path.join(barcode_dir, b + ext)] for b in bins_of_piece), dtype=np.int32,
                                     count=len(bins_of_piece))
                else:
                    if not paths:
                        paths.append(target)
                        file_pos = np.zeros(1, dtype=np.int64)
                    pf = np.zeros(pr.size, dtype=np.int32)
                if pr.size:
                    t1 = time.perf_counter()
                    _emit(rs_, pr, ps_, pn_, num, pf, paths, fastq, file_pos, gz)
                    busy["write_call"] = busy.get("write_call", 0.0) + time.perf_counter() - t1
                for k in np.unique(pf) if pr.size else []:
                    sel = pf == k
                    rr = np.unique(pr[sel])
                    n0, b0 = stats.get(paths[k], (0, 0))
                    if barcode_dir is not None:
                        # the reference counts reads (not pieces) and their end-trimmed (or whole) lengths
                        stats[paths[k]] = (n0 + int(rr.size), b0 + int((rs_.lengths[rr] if whole else tlen[rr]).sum()))
                    else:
                        stats[paths[k]] = (n0 + int(rr.size), b0 + int(pn_[sel].sum()))
                t1 = time.perf_counter()
                rs_.close()
                busy["free"] = busy.get("free", 0.0) + time.perf_counter() - t1
                busy["write"] += time.perf_counter() - t0
        except BaseException as e:           # noqa
            failure.append(e)
            while to_write.get() is not None:
                pass

    lt = threading.Thread(target=loader, daemon=True)
    wt = threading.Thread(target=writer, daemon=True)
    lt.start(); wt.start()
    st_all, et_all, calls_all = [], [], []
    matching = match_idx = orientation = None
    try:
        rs = first
        while rs is not None:
            if failure:
                break
            t0 = time.perf_counter()
            R = rs.count
            reads = None
            if R:
                reads = DeviceReads(torch.from_numpy(rs.arena).to(dev), torch.from_numpy(rs.offsets.copy()).to(dev),
                                    torch.from_numpy(rs.lengths.copy()).to(dev))
            if matching is None:
                check_idx = np.arange(min(R, max(0, opts.check_reads)), dtype=np.int64)
                matching, match_idx, orientation = _find_sets(pl, panel, reads, check_idx, opts, barcode_dir)
                res.matching_sets = [s.name for s in matching]
                res.barcode_orientation = orientation
            start_trim, end_trim, ci, hits, _ = _scan_reads(pl, reads, R, match_idx, opts, barcode_dir, orientation)
            if hasattr(pl.aligner, "sync"):
                pl.aligner.sync()
            if hits is not None and hits.read.numel():
                h = torch.stack([hits.read, hits.adapter.to(torch.int64), hits.start.to(torch.int64),
                                 hits.end.to(torch.int64)], dim=1).cpu().numpy()
            else:
                h = np.zeros((0, 4), dtype=np.int64)
            st, et = start_trim.cpu().numpy(), end_trim.cpu().numpy()
            calls = None
            if barcode_dir is not None:
                names_all = _barcode_bin_names(pl, match_idx, orientation)
                calls = [names_all[k] if k >= 0 else "none" for k in ci]
                calls_all.extend(calls)
            pr, ps_, pn_, num, tlen, n_split = _plan_pieces(opts, rs.lengths, st, et, h, calls, barcode_dir, discard_middle,
                                                           matching, pl, match_idx)
            res.middle_hit_reads += n_split
            res.n_reads += R
            st_all.append(st); et_all.append(et)
            bins_of_piece = [calls[r] for r in pr] if barcode_dir is not None else None
            del reads
            busy["scan"] += time.perf_counter() - t0
            to_write.put((rs, pr, ps_, pn_, num, bins_of_piece, tlen))
            rs = loaded.get()
    finally:
        stop.set()
        to_write.put(None)
        wt.join()
        while lt.is_alive() or not loaded.empty():   # blocks the loader had in flight when we stopped early
            try:
                x = loaded.get(timeout=0.05)
                if x is not None:
                    x.close()
            except queue.Empty:
                pass
        if aligner is None:
            pl.close()
        if gz_in is not None:
            gz_in.close()
    if failure:
        # a streamed run that fails part-way (a later block that cannot be parsed, a full disk) has already written the
        # earlier blocks: a whole-file run would have written nothing, so nothing is left behind here either
        for path in list(paths):
            try:
                if path and path != "-" and os.path.isfile(path):
                    os.remove(path)
            except OSError:
                pass
        raise failure[0]
    res.start_trim = np.concatenate(st_all) if st_all else np.zeros(0, dtype=np.int32)
    res.end_trim = np.concatenate(et_all) if et_all else np.zeros(0, dtype=np.int32)
    res.barcode_calls = calls_all if barcode_dir is not None else None
    # ---- what a whole-file run does after writing -------------------------------------------------------
    if barcode_dir is not None:
        for path in paths:
            res.files[path] = stats.get(path, (0, 0))
            if gz:
                gz_finish(path)
    elif output is not None:
        if not paths or file_pos[0] == 0:
            open(target, "wb").close()                              # the reference always creates the file
        if gz:
            gz_finish(target)
        res.files[output] = stats.get(target, (0, 0))
    res.seconds = {"wall": time.perf_counter() - t_start, "load_busy": busy["load"], "scan_busy": busy["scan"], "write_busy": busy["write"],
                   "write_call_busy": busy.get("write_call", 0.0), "free_busy": busy.get("free", 0.0)}
    return res


def run_sharded(input_path, output, barcode_dir, opts: Options, device=None, aligner=None,
                adapter_panel: List[AdapterSet] = None) -> Optional[RunResult]:
    """One plain FASTQ file -- or one gzip file of sized members, addressed by its inflated bytes -- over the ranks of a
    torch.distributed job WITHOUT any rank touching the whole file: rank r
    parses the records that start in its W-th of the file's bytes (pc_fastq_find_record / pc_readset_load_segment),
    scans them on its GPU and writes its own span of the shared output files (pc_readset_write_sizes, exchanged, give
    every rank its positions; pc_readset_write_shared).  The collectives: phase A's presence table (MAX), read counts,
    output sizes, the names of the bins in use.  The files are the single-process ones byte for byte -- Porechop's
    decisions are per read once the presence table is known (porechop.py:224-273,607-734).
    -> None when this route does not apply on EVERY rank (input not a streamable plain FASTQ file; output to stdout; the
    ranks were given different paths): run() then falls back to every rank loading the input and rank 0 writing."""
    import torch.distributed as dist
    from .distributed import all_agree, all_gather_ints, all_gather_objects
    from .io import fastq_record_start
    rank, world = dist.get_rank(), dist.get_world_size()
    t_start = time.perf_counter()
    target_dir_or_file = os.path.abspath(barcode_dir if barcode_dir is not None else output) if (barcode_dir or output) else None
    ok = os.path.isfile(input_path) and target_dir_or_file is not None
    # the same files on every rank?  (tests run the ranks on private copies: those runs gather to rank 0 as before)
    seen = all_gather_objects((os.path.abspath(input_path), target_dir_or_file, os.path.getsize(input_path) if ok else -1))
    ok = ok and all(x == seen[0] for x in seen)
    rs, b0, b1 = None, 0, 0
    if ok and _is_gzip(input_path):
        # a gzip file of SIZED members (this package's own output; bgzip): positions are those of the inflated bytes, a rank
        # inflates only the members that hold its records (pc_gz_sized_find_record / pc_readset_load_gz_range); any other
        # gzip file cannot be cut without inflating all of it: not this route
        from .io import gz_member_start, gz_sized_record_start, gz_sized_size
        size = gz_sized_size(input_path)
        if not size:
            # an ordinary multi-member gzip file (`cat *.fastq.gz`): cut at MEMBER starts of the compressed bytes -- rank r takes
            # the members that start in its W-th (pc_gz_member_start validates a start by inflating the member behind it) and
            # inflates them with its own cores (GzStream over that byte range); the members must hold whole records (they do when
            # they were files).  ONE member: rank 0 gets all of it, the route still applies (the other ranks write nothing).
            csize = os.path.getsize(input_path)
            b0 = 0 if rank == 0 else gz_member_start(input_path, csize * rank // world)
            b1 = csize if rank == world - 1 else gz_member_start(input_path, csize * (rank + 1) // world)
            if b0 is None or b1 is None:
                ok = False
            elif b1 > b0:
                try:
                    gzs = GzStream(input_path, b0, b1)
                    got = gzs.next(1 << 62, 0)
                    gzs.close()
                except ValueError:
                    got = False
                if got is False:
                    ok = False
                else:
                    rs = got                      # (None: members without a record)
        else:
            b0 = 0 if rank == 0 else gz_sized_record_start(input_path, size * rank // world)
            b1 = size if rank == world - 1 else gz_sized_record_start(input_path, size * (rank + 1) // world)
            if b0 is None or b1 is None:
                ok = False
            elif b1 > b0:
                rs = ReadSet.gz_range(input_path, b0, b1)
                if rs is None:
                    ok = False
    elif ok:
        size = os.path.getsize(input_path)
        b0 = 0 if rank == 0 else fastq_record_start(input_path, size * rank // world)
        b1 = size if rank == world - 1 else fastq_record_start(input_path, size * (rank + 1) // world)
        if b0 is None or b1 is None:
            ok = False
        elif b1 > b0:
            rs, nxt = ReadSet.segment(input_path, b0, b1 - b0)
            if rs is None or nxt != b1:
                ok = False
    if not all_agree(ok, device if aligner is None else None):
        if rs is not None:
            rs.close()
        return None
    discard_middle = opts.discard_middle or barcode_dir is not None
    R = rs.count if rs is not None else 0
    counts = all_gather_ints([R], device if aligner is None else None)[:, 0].numpy()
    first_read, total = int(counts[:rank].sum()), int(counts.sum())
    # (the type of the file is the type of its first record: rank 0's share always holds it)
    kinds = all_gather_objects(None if rs is None else bool(rs.is_fastq))
    is_fastq = next((k for k in kinds if k is not None), True)
    res = RunResult(n_reads=total, read_type="FASTQ" if is_fastq else "FASTA")
    res.seconds["load"] = time.perf_counter() - t_start
    check_idx = np.arange(max(0, min(R, max(0, opts.check_reads) - first_read)), dtype=np.int64)

    panel = list(adapter_panel) if adapter_panel is not None else panel_rules.load_panel()
    params = ScanParams(end_size=opts.end_size, min_trim_size=opts.min_trim_size, extra_end_trim=opts.extra_end_trim,
                        end_threshold=opts.end_threshold, middle_threshold=opts.middle_threshold,
                        adapter_threshold=opts.adapter_threshold, check_reads=opts.check_reads,
                        scores=tuple(int(x) for x in opts.scoring_scheme))
    pl = Pipeline(panel, params, device=device, aligner=aligner)
    dev = pl.device
    if aligner is None:
        pl.aligner.lib.pc_jit_async(1)
    coll_dev = dev if aligner is None else None
    try:
        t0 = time.perf_counter()
        reads = None
        if R:
            reads = DeviceReads(torch.from_numpy(rs.arena).to(dev), torch.from_numpy(rs.offsets.copy()).to(dev),
                                torch.from_numpy(rs.lengths.copy()).to(dev))
        matching, match_idx, orientation = _find_sets(pl, panel, reads, check_idx, opts, barcode_dir, sharded=True)
        res.matching_sets = [s.name for s in matching]
        res.barcode_orientation = orientation
        start_trim, end_trim, ci, hits, _ = _scan_reads(pl, reads, R, match_idx, opts, barcode_dir, orientation)
        if hasattr(pl.aligner, "sync"):
            pl.aligner.sync()
        if hits is not None and hits.read.numel():
            h = torch.stack([hits.read, hits.adapter.to(torch.int64), hits.start.to(torch.int64),
                             hits.end.to(torch.int64)], dim=1).cpu().numpy()
        else:
            h = np.zeros((0, 4), dtype=np.int64)
        st, et = start_trim.cpu().numpy(), end_trim.cpu().numpy()
        calls = None
        if barcode_dir is not None:
            names_all = _barcode_bin_names(pl, match_idx, orientation)
            calls = [names_all[k] if k >= 0 else "none" for k in ci]
        res.start_trim, res.end_trim, res.barcode_calls = st, et, calls          # this rank's reads
        res.first_read, res.local_reads = first_read, R
        res.seconds["scan"] = time.perf_counter() - t0

        # ---- this rank's pieces, the files they go to, and where ------------------------------------------
        t0 = time.perf_counter()
        fmt, gz = _resolve_format(opts, output, barcode_dir, res.read_type, input_path)
        res.out_format = fmt
        fastq = fmt != "fasta"
        lengths = rs.lengths if R else np.zeros(0, dtype=np.int32)
        pr, ps_, pn_, num, tlen, n_split = _plan_pieces(opts, lengths, st, et, h, calls, barcode_dir, discard_middle,
                                                       matching, pl, match_idx)
        tallies = all_gather_ints([n_split, int((st > 0).sum()), int((et > 0).sum())], coll_dev).sum(dim=0)
        res.middle_hit_reads = int(tallies[0])
        res.counts = {"start_trimmed": int(tallies[1]), "end_trimmed": int(tallies[2])}
        if barcode_dir is not None:
            mine = sorted({calls[r] for r in pr.tolist()})
            bins = sorted(set().union(*all_gather_objects(mine)))
            index = {b: k for k, b in enumerate(bins)}
            paths = [os.path.join(barcode_dir, b + "." + fmt) for b in bins]
            pf = np.fromiter((index[calls[r]] for r in pr.tolist()), dtype=np.int32, count=int(pr.size))
            paths = [p + (".gz" if gz else "") for p in paths]
        else:
            paths = [output]
            pf = np.zeros(pr.size, dtype=np.int32)
        finals = paths
        nf = len(paths)
        # gz: every rank deflates its own pieces in memory first (independent members: any concatenation of them is a valid
        # file), the COMPRESSED sizes are what the ranks exchange, and each writes its image at its position
        img = None
        failed = None
        try:
            if gz and R and nf:
                img = rs.compress(pr, ps_, pn_, num, pf, nf, fastq)
                sizes = img.sizes()[0]
            else:
                sizes = rs.write_sizes(pr, ps_, pn_, num, pf, nf, fastq) if (R and nf) else np.zeros(nf, dtype=np.int64)
        except OSError as e:
            failed, sizes = e, np.zeros(nf, dtype=np.int64)
        # per file: bytes, reads and bases of every rank (the reference counts reads, not pieces, and for bins their
        # end-trimmed -- or whole -- lengths)
        per_file = np.zeros((nf, 3), dtype=np.int64)
        per_file[:, 0] = sizes
        for k in range(nf):
            sel = pf == k
            rr = np.unique(pr[sel])
            per_file[k, 1] = rr.size
            per_file[k, 2] = int((lengths[rr] if opts.untrimmed else tlen[rr]).sum()) if barcode_dir is not None else int(pn_[sel].sum())
        everyone = all_gather_ints(per_file.reshape(-1), coll_dev).numpy().reshape(world, nf, 3) if nf else np.zeros((world, 0, 3), dtype=np.int64)
        pos = everyone[:rank, :, 0].sum(axis=0).astype(np.int64)
        if rank == 0:
            if barcode_dir is not None:
                os.makedirs(barcode_dir, exist_ok=True)
            for path in paths:
                open(path, "wb").close()                 # created / truncated ONCE, before any rank writes its span
        dist.barrier()
        # a rank that fails here alone (a full disk, a vanished directory) must not leave the others waiting in the
        # next collective: the outcome is exchanged, every rank raises, rank 0 removes the partial files
        try:
            if failed is None and R and pr.size:
                if img is not None:
                    img.write(paths, np.ascontiguousarray(pos), shared=True)
                else:
                    rs.write_shared(pr, ps_, pn_, num, pf, paths, fastq, np.ascontiguousarray(pos))
        except OSError as e:
            failed = e
        finally:
            if img is not None:
                img.close()
        totals = everyone.sum(axis=0)
        if not all_agree(failed is None, coll_dev):
            if rank == 0:
                for path in paths:
                    try:
                        os.remove(path)
                    except OSError:
                        pass
            raise failed if failed is not None else OSError("Error: could not write the output reads (another rank failed)")
        if rank == 0 and gz:
            for path in paths:
                gz_finish(path)
        for k in range(nf):
            res.files[finals[k]] = (int(totals[k, 1]), int(totals[k, 2]))
        dist.barrier()
        res.seconds["write"] = time.perf_counter() - t0
        res.seconds["wall"] = time.perf_counter() - t_start
        return res
    finally:
        if aligner is None:
            pl.close()
        if rs is not None:
            rs.close()


def run(input_path, output=None, barcode_dir=None, options: Options = None, device=None, aligner=None,
        adapter_panel: List[AdapterSet] = None) -> RunResult:
    """Porechop's main() on arrays.  output=None and barcode_dir=None writes to stdout.
    `aligner` is for tests only (see Pipeline).

    Under torch.distributed (one process per GPU) the reads are sharded over the ranks and the adapter-set presence
    table of phase A is MAX-all-reduced (the only cross-read quantity in Porechop).  A plain FASTQ or FASTA file, a gzip
    file of sized members and a `cat` of gzip members take run_sharded: every rank parses, scans and WRITES only its own
    byte range of the input / span of the shared output files, and its RunResult holds its own reads' trims (first_read,
    local_reads say which).  Anything else (one big gzip member, a directory, stdout) falls back to every rank loading the input, contiguous blocks of about equal bases per
    rank, the per-read results gathered in rank order and rank 0 alone planning and writing.  Either way the files
    are the single-process ones."""
    opts = options or Options()
    if len(tuple(opts.scoring_scheme)) != 4:
        raise UsageError("Error: incorrectly formatted scoring scheme")
    if barcode_dir is not None and output is not None:
        raise UsageError("Error: only one of the following options may be used: --output, --barcode_dir")
    if opts.untrimmed and barcode_dir is None:
        raise UsageError("Error: --untrimmed can only be used with --barcode_dir")
    discard_middle = opts.discard_middle or barcode_dir is not None          # porechop.py:203-204
    input_path = str(input_path)

    import torch.distributed as dist
    # (a .gz file holds about three times its size in FASTQ)
    if (os.path.isfile(input_path) and os.path.getsize(input_path) * (3 if _is_gzip(input_path) else 1) > 2 * _stream_block_bytes()
            and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)):
        streamed = run_streamed(input_path, output, barcode_dir, opts, device=device, aligner=aligner, adapter_panel=adapter_panel)
        if streamed is not None:
            return streamed

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        shared = run_sharded(input_path, output, barcode_dir, opts, device=device, aligner=aligner, adapter_panel=adapter_panel)
        if shared is not None:
            return shared

    t_last = [time.perf_counter()]

    def lap(stage, sync=False):
        if sync and aligner is None and torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        res.seconds[stage] = res.seconds.get(stage, 0.0) + now - t_last[0]
        t_last[0] = now

    rs, check_idx, albacore = _load(input_path, opts.check_reads)
    res = RunResult(n_reads=rs.count, read_type="FASTQ" if rs.is_fastq else "FASTA")
    R_all = rs.count
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if sharded else (0, 1)
    lo_r, hi_r = shard_by_bases(rs.lengths, world, rank)
    R = hi_r - lo_r                                             # reads this rank scans
    check_idx = check_idx[(check_idx >= lo_r) & (check_idx < hi_r)] - lo_r
    lap("load")

    panel = list(adapter_panel) if adapter_panel is not None else panel_rules.load_panel()
    params = ScanParams(end_size=opts.end_size, min_trim_size=opts.min_trim_size, extra_end_trim=opts.extra_end_trim,
                        end_threshold=opts.end_threshold, middle_threshold=opts.middle_threshold,
                        adapter_threshold=opts.adapter_threshold, check_reads=opts.check_reads,
                        scores=tuple(int(x) for x in opts.scoring_scheme))
    pl = Pipeline(panel, params, device=device, aligner=aligner)
    dev = pl.device
    if aligner is None:
        # a one-shot run should not wait for hiprtc: specialised kernels are compiled on a worker
        # thread and picked up by later launches (pc_jit_async, include/porechop_amd.h)
        pl.aligner.lib.pc_jit_async(1)
    try:
        reads = None
        if R:
            a0 = int(rs.offsets[lo_r])
            a1 = min(int(rs.offsets[hi_r - 1]) + int(rs.lengths[hi_r - 1]) + 64, rs.arena.size)   # >= 16 readable bytes past the end (the kernels fetch 16 columns per load)
            reads = DeviceReads(torch.from_numpy(rs.arena[a0:a1]).to(dev), torch.from_numpy(rs.offsets[lo_r:hi_r] - a0).to(dev),
                                torch.from_numpy(rs.lengths[lo_r:hi_r].copy()).to(dev))
        lap("upload", sync=True)

        # ---- phase A and the set-level rules ---------------------------------------------
        matching, match_idx, orientation = _find_sets(pl, panel, reads, check_idx if R else np.zeros(0, dtype=np.int64), opts,
                                                      barcode_dir, sharded)
        res.matching_sets = [s.name for s in matching]
        res.barcode_orientation = orientation
        lap("phase_a", sync=True)

        calls = None
        start_trim, end_trim, ci, hits, names = _scan_reads(pl, reads, R, match_idx, opts, barcode_dir, orientation, lap)
        if hasattr(pl.aligner, "sync"):
            pl.aligner.sync()

        # ---- per-read results of all ranks, in read order --------------------------------
        if hits is not None and hits.read.numel():
            h = torch.stack([hits.read + lo_r, hits.adapter.to(torch.int64), hits.start.to(torch.int64),
                             hits.end.to(torch.int64)], dim=1)
        else:
            h = torch.zeros((0, 4), dtype=torch.int64, device=dev)
        ci_t = torch.from_numpy(ci).to(dev)
        if sharded:
            start_trim, end_trim, ci_t, h = (gather_in_order(x) for x in (start_trim, end_trim, ci_t, h))
            lap("gather", sync=True)
            if rank != 0:
                res.start_trim, res.end_trim = start_trim.cpu().numpy(), end_trim.cpu().numpy()
                return res                                       # rank 0 writes
        R = R_all
        st = start_trim.cpu().numpy()
        et = end_trim.cpu().numpy()
        h = h.cpu().numpy()
        if barcode_dir is not None:
            names_all = _barcode_bin_names(pl, match_idx, orientation)     # the same list on every rank
            calls = [names_all[k] if k >= 0 else "none" for k in ci_t.cpu().numpy()]
            if albacore is not None:                               # nanopore_read.py:468-473
                calls = [c if (a is None or a == c) else "none" for c, a in zip(calls, albacore)]
        res.start_trim, res.end_trim, res.barcode_calls = st, et, calls

        # ---- which pieces of which reads -------------------------------------------------
        fmt, gz = _resolve_format(opts, output, barcode_dir, res.read_type, input_path)
        res.out_format = fmt
        whole = opts.untrimmed
        pr, ps_, pn_, num, tlen, n_split = _plan_pieces(opts, rs.lengths, st, et, h, calls, barcode_dir, discard_middle,
                                                       matching, pl, match_idx)
        res.middle_hit_reads = n_split

        lap("plan_output")
        # ---- write -------------------------------------------------------------------------
        fastq = fmt != "fasta"                                      # porechop.py:667,711,727
        if barcode_dir is not None:
            os.makedirs(barcode_dir, exist_ok=True)
            all_bins = sorted(set(calls))
            lookup = {b: k for k, b in enumerate(all_bins)}
            bin_of_read = np.fromiter((lookup[c] for c in calls), dtype=np.int32, count=len(calls))
            used = np.unique(bin_of_read[pr]) if pr.size else np.zeros(0, dtype=np.int32)
            bins = [all_bins[k] for k in used]
            remap = np.full(len(all_bins), -1, dtype=np.int32)
            remap[used] = np.arange(used.size, dtype=np.int32)
            pf = remap[bin_of_read[pr]] if pr.size else np.zeros(0, dtype=np.int32)
            paths = [os.path.join(barcode_dir, b + "." + fmt + (".gz" if gz else "")) for b in bins]
            _emit(rs, pr, ps_, pn_, num, pf, paths, fastq, np.zeros(len(paths), dtype=np.int64), gz)
            for k, (b, path) in enumerate(zip(bins, paths)):
                sel = pf == k
                # the reference counts reads (not pieces) and their end-trimmed (or whole) lengths
                rr = np.unique(pr[sel])
                res.files[path] = (int(rr.size), int((rs.lengths[rr] if whole else tlen[rr]).sum()))
                if gz:
                    gz_finish(path)
        elif output is None:
            rs.write(pr, ps_, pn_, num, np.zeros(pr.size, dtype=np.int32), ["-"], fastq)
        else:
            open(output, "wb").close()                              # the reference always creates the file
            if pr.size:
                _emit(rs, pr, ps_, pn_, num, np.zeros(pr.size, dtype=np.int32), [output], fastq, np.zeros(1, dtype=np.int64), gz)
            if gz:
                gz_finish(output)
            res.files[output] = (int(np.unique(pr).size), int(pn_.sum()))
        lap("write")
        return res
    finally:
        if aligner is None:
            pl.close()
        rs.close()
