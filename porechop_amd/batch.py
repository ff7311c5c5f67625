"""Batch interface over the C ABI: many (window, adapter) pairs per call.

Host arrays are numpy; device-resident inputs are torch tensors (PyTorch is used only to own
HBM buffers and streams -- the compute is the HIP library).
"""
import ctypes

import numpy as np

from ._lib import check, load_library

RESULT_INTS = 8      # readStart, readEnd, adapterStart, adapterEnd, rawScore, matches, alignedLen, fullLen
MODE_AUTO, MODE_TRACE, MODE_TWO_PASS, MODE_SCORE, MODE_TRACE_AT = 0, 1, 2, 3, 4
DEFAULT_SCORES = (3, -6, -5, -2)   # porechop/porechop.py:145
INT_MIN = -2147483648


def format_result(rec):
    """One 8-int record -> the reference's 7-field string (formatted by the C side, i.e. by the
    same libc printf the reference uses through std::to_string)."""
    lib = load_library()
    arr = (ctypes.c_int32 * RESULT_INTS)(*[int(x) for x in rec])
    buf = ctypes.create_string_buffer(160)
    lib.pc_format_result(arr, buf, 160)
    return buf.value.decode()


def format_results(recs):
    """[n, 8] int32 records -> list of the reference's 7-field strings (pc_format_results: one call for all)."""
    recs = np.ascontiguousarray(recs, dtype=np.int32).reshape(-1, RESULT_INTS)
    n = int(recs.shape[0])
    if n == 0:
        return []
    lib = load_library()
    buf = ctypes.create_string_buffer(161 * n)
    used = ctypes.c_int64()
    check(lib.pc_format_results(recs.ctypes.data, n, buf, 161 * n, ctypes.byref(used)), "pc_format_results")
    out = buf.raw[:used.value].decode("ascii").split("\n")
    out.pop()
    return out


def records_to_fields(recs):
    """[n,8] int32 -> what porechop/nanopore_read.py:476-491 (align_adapter) derives per call:
    full_adapter_identity, aligned_identity (float64), read_start, read_end (= field1 + 1)."""
    recs = np.asarray(recs)
    failed = recs[:, 0] == -1
    with np.errstate(divide="ignore", invalid="ignore"):
        m = recs[:, 5].astype(np.float64)
        aligned = 100.0 * m / recs[:, 6].astype(np.float64)
        full = 100.0 * m / recs[:, 7].astype(np.float64)
    read_start = recs[:, 0].copy()
    read_end = recs[:, 1] + 1
    full[failed] = 0.0
    aligned[failed] = 0.0
    read_end[failed] = 0
    return full, aligned, read_start, read_end


class Aligner:
    """One GPU context: a scoring scheme + an adapter panel + scratch buffers."""
    fast_prefilter = True      # prefilter_rows is the device's exact prefilter (callers may put it in front of the middle scan)
    score_end_cell = True      # MODE_SCORE records carry the reference's end cell (row, column) besides the score
    reduce_takes_mask = True   # phase_b_reduce(traced_mask=...): pairs whose bit is clear are skipped without a load
    trace_at = True            # MODE_TRACE_AT: the traced record of a pair whose MODE_SCORE record (its end cell) is known

    def __init__(self, adapters, scores=DEFAULT_SCORES, device=-1):
        self.lib = load_library()
        self._ctx = ctypes.c_void_p()
        check(self.lib.pc_create(ctypes.byref(self._ctx), device), "pc_create")
        self.scores = tuple(int(s) for s in scores)
        check(self.lib.pc_set_scores(self._ctx, *self.scores), "pc_set_scores")
        self.set_adapters(adapters)

    def set_adapters(self, adapters):
        """Replace the adapter table (indices used by later calls refer to the new list)."""
        self.adapters = [a if isinstance(a, bytes) else a.encode() for a in adapters]
        arr = (ctypes.c_char_p * max(1, len(self.adapters)))(*self.adapters)
        check(self.lib.pc_set_adapters(self._ctx, arr, len(self.adapters)), "pc_set_adapters")

    def close(self):
        if self._ctx:
            self.lib.pc_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host buffers ------------------------------------------------------------------
    def align_host(self, arena, win_off, win_len, adapter_idx, mode=MODE_AUTO):
        """arena: bytes / uint8 array; win_off int64[n]; win_len int32[n]; adapter_idx int32[n]
        -> int32 [n, 8]"""
        arena = np.frombuffer(arena, dtype=np.uint8) if isinstance(arena, (bytes, bytearray)) \
            else np.ascontiguousarray(arena, dtype=np.uint8)
        win_off = np.ascontiguousarray(win_off, dtype=np.int64)
        win_len = np.ascontiguousarray(win_len, dtype=np.int32)
        adapter_idx = np.ascontiguousarray(adapter_idx, dtype=np.int32)
        n = win_off.shape[0]
        out = np.zeros((n, RESULT_INTS), dtype=np.int32)
        check(self.lib.pc_align_batch_host(self._ctx, arena.ctypes.data, arena.size, win_off.ctypes.data,
                                           win_len.ctypes.data, adapter_idx.ctypes.data, n, mode,
                                           out.ctypes.data), "pc_align_batch_host")
        return out

    def align_pairs(self, pairs, mode=MODE_AUTO):
        """pairs: iterable of (read_str, adapter_index) -> int32 [n, 8]"""
        offs, lens, idx, chunks, pos = [], [], [], [], 0
        for rd, ai in pairs:
            b = rd if isinstance(rd, bytes) else rd.encode()
            offs.append(pos)
            lens.append(len(b))
            idx.append(ai)
            chunks.append(b)
            pos += len(b)
        return self.align_host(b"".join(chunks), offs, lens, idx, mode)

    # ---- device buffers (torch tensors on the GPU) --------------------------------------
    def scan_device(self, arena, win_off, win_len, job_adapter, job_start, max_len, out,
                    mode=MODE_AUTO, stream=None, job_adapter_b=None):
        """arena uint8[*], win_off int64[n], win_len int32[n]: CUDA(HIP) tensors describing n windows.
        job_adapter int32[k], job_start int64[k+1] (window ranges), optional job_adapter_b int32[k]
        (-1 = none): host numpy.  out int32[total,8] with total = sum n_k * (1 or 2), job order,
        adapter A's records before adapter B's.  Asynchronous; call sync()."""
        import torch
        assert arena.is_cuda and win_off.is_cuda and win_len.is_cuda and out.is_cuda
        assert win_off.dtype == torch.int64 and win_len.dtype == torch.int32 and out.dtype == torch.int32
        job_adapter = np.ascontiguousarray(job_adapter, dtype=np.int32)
        job_start = np.ascontiguousarray(job_start, dtype=np.int64)
        jb = None
        if job_adapter_b is not None:
            jb = np.ascontiguousarray(job_adapter_b, dtype=np.int32)
        n = win_off.shape[0]
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_scan_device(self._ctx, arena.data_ptr(), win_off.data_ptr(), win_len.data_ptr(), n,
                                      job_adapter.ctypes.data, jb.ctypes.data if jb is not None else None,
                                      job_start.ctypes.data, len(job_adapter),
                                      int(max_len), mode, out.data_ptr(), ctypes.c_void_p(s)),
              "pc_scan_device")

    def set_length_hint(self, typical_len):
        """Typical window length of the following whole-read scans (0 = about uniform): load balancing of the
        score pass only, never results (pc_set_length_hint)."""
        check(self.lib.pc_set_length_hint(self._ctx, int(max(0, typical_len))), "pc_set_length_hint")

    def set_int16_only(self, enabled=True):
        """Use the packed-int16 kernel variants even where the packed-fp16 ones are proven exact (cross-checks)."""
        check(self.lib.pc_set_int16_only(self._ctx, 1 if enabled else 0), "pc_set_int16_only")

    def set_timing(self, enabled=True):
        check(self.lib.pc_set_timing(self._ctx, 1 if enabled else 0), "pc_set_timing")

    def get_timing(self, stream=None):
        """-> dict kind -> (ms, launches, pairs) since the last call, for the kinds 'score' (generic
        score-only scan), 'plan', 'trace', 'score_spec' (run-time specialised score-only scan) and 'prefilter'
        (exact prefilter, all launches of a call as one region; its "pairs" are (window, adapter) pairs) and 'seed_scan' (the
        prefilter's seed scan alone: a sub-interval of 'prefilter'; its "pairs" are windows) and 'select' (the selection
        kernels of phase B's exact pruning)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        ms = (ctypes.c_double * 7)()
        ln = (ctypes.c_int64 * 7)()
        pr = (ctypes.c_int64 * 7)()
        check(self.lib.pc_get_timing(self._ctx, ctypes.c_void_p(s), ms, ln, pr), "pc_get_timing")
        return {k: (ms[i], ln[i], pr[i]) for i, k in enumerate(("score", "plan", "trace", "score_spec", "prefilter", "seed_scan", "select"))}

    def phase_b_reduce(self, records, n, job_record_offset, job_side, end_size, min_trim_size, extra_end_trim,
                       end_threshold, start_trim, end_trim, bins=None, barcode_threshold=0.0, barcode_diff=0.0,
                       require_two=False, call=None, stream=None, traced_mask=None):
        """Per-read reduction of end-window records on the device (pc_phase_b_reduce): records int32[*,8]
        written by scan_device for len(job_side) jobs over the same n reads; start_trim / end_trim (and call,
        when bins = [(start job or -1, end job or -1), ...] is given) are int32[n] CUDA tensors."""
        import torch
        assert records.is_cuda and start_trim.is_cuda and end_trim.is_cuda
        assert records.dtype == torch.int32 and start_trim.dtype == torch.int32 and end_trim.dtype == torch.int32
        off = np.ascontiguousarray(job_record_offset, dtype=np.int64)
        side = np.ascontiguousarray(job_side, dtype=np.int32)
        nb = 0 if bins is None else len(bins)
        bs = np.ascontiguousarray([b[0] for b in bins] if nb else [0], dtype=np.int32)
        be = np.ascontiguousarray([b[1] for b in bins] if nb else [0], dtype=np.int32)
        if nb:
            assert call is not None and call.is_cuda and call.dtype == torch.int32
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if traced_mask is not None:
            # traced_mask int64 [njobs, (n + 63) // 64]: the union of the selection rounds' masks (pc_phase_b_reduce_masked)
            assert traced_mask.is_cuda and traced_mask.dtype == torch.int64 and traced_mask.is_contiguous()
            assert tuple(traced_mask.shape) == (len(side), (int(n) + 63) // 64)
        check(self.lib.pc_phase_b_reduce_masked(self._ctx, records.data_ptr(), int(n), len(side), off.ctypes.data, side.ctypes.data,
                                                int(end_size), int(min_trim_size), int(extra_end_trim), float(end_threshold),
                                                start_trim.data_ptr(), end_trim.data_ptr(), nb, bs.ctypes.data, be.ctypes.data,
                                                float(barcode_threshold), float(barcode_diff), 1 if require_two else 0,
                                                call.data_ptr() if nb else None,
                                                traced_mask.data_ptr() if traced_mask is not None else None, ctypes.c_void_p(s)),
              "pc_phase_b_reduce")

    def phase_b_select(self, records, n, job_off, job_side, job_len, job_calls, start_len, end_len, end_size, min_trim_size,
                       extra_end_trim, end_threshold, rnd, call_level, call_level_diff, mask_out, counts, mask_prev=None,
                       start_trim=None, end_trim=None, best_full=None, ub_trim_out=None, ub_full_out=None, stream=None):
        """Exact pruning of phase B, selection (pc_phase_b_select): all arguments CUDA tensors -- job_off int64[J],
        job_side / job_len / job_calls int32[J], start_len / end_len int32[n], mask_out / mask_prev int64[J, (n+63)//64],
        counts int64[J], best_full float64[2, n]."""
        import torch
        J = int(job_side.shape[0])
        assert records.dtype == torch.int32 and job_off.dtype == torch.int64 and mask_out.dtype == torch.int64 and counts.dtype == torch.int64
        assert job_side.dtype == job_len.dtype == job_calls.dtype == start_len.dtype == end_len.dtype == torch.int32
        assert tuple(mask_out.shape) == (J, (int(n) + 63) // 64) and mask_out.is_contiguous() and int(counts.shape[0]) == J
        ptr = lambda t: None if t is None else t.data_ptr()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_phase_b_select(self._ctx, records.data_ptr(), int(n), J, job_off.data_ptr(), job_side.data_ptr(),
                                         job_len.data_ptr(), job_calls.data_ptr(), start_len.data_ptr(), end_len.data_ptr(),
                                         int(end_size), int(min_trim_size), int(extra_end_trim), float(end_threshold), int(rnd),
                                         float(call_level), float(call_level_diff), ptr(mask_prev), ptr(start_trim), ptr(end_trim),
                                         ptr(best_full), mask_out.data_ptr(), counts.data_ptr(), ptr(ub_trim_out), ptr(ub_full_out),
                                         ctypes.c_void_p(s)), "pc_phase_b_select")

    def phase_b_gather(self, mask, n, job_first, cursor, job_off, job_side, start_off, start_len, end_off, end_len,
                       win_off, win_len, dest, pair_job, pair_read, stream=None):
        """The selected pairs as the window lists of a traced scan (pc_phase_b_gather)."""
        import torch
        J = int(job_side.shape[0])
        assert mask.dtype == torch.int64 and job_first.dtype == torch.int64 and cursor.dtype == torch.int64
        assert start_off.dtype == end_off.dtype == win_off.dtype == dest.dtype == pair_read.dtype == torch.int64
        assert start_len.dtype == end_len.dtype == win_len.dtype == pair_job.dtype == torch.int32
        assert start_off.is_contiguous() and end_off.is_contiguous() and start_len.is_contiguous() and end_len.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_phase_b_gather(self._ctx, mask.data_ptr(), int(n), J, job_first.data_ptr(), cursor.data_ptr(),
                                         job_off.data_ptr(), job_side.data_ptr(), start_off.data_ptr(), start_len.data_ptr(),
                                         end_off.data_ptr(), end_len.data_ptr(), win_off.data_ptr(), win_len.data_ptr(),
                                         dest.data_ptr(), pair_job.data_ptr(), pair_read.data_ptr(), ctypes.c_void_p(s)),
              "pc_phase_b_gather")

    def gather_records(self, records, index, out=None, stream=None):
        """out[k] = records[index[k]] on the device (pc_gather_records); records int32 [*, 8], index int64 [count]."""
        import torch
        assert records.dtype == torch.int32 and records.is_contiguous() and index.dtype == torch.int64 and index.is_contiguous()
        n = int(index.shape[0])
        if out is None:
            out = torch.empty((n, RESULT_INTS), dtype=torch.int32, device=records.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_gather_records(self._ctx, records.data_ptr(), index.data_ptr(), n, out.data_ptr(), ctypes.c_void_p(s)),
              "pc_gather_records")
        return out

    def phase_b_scatter(self, traced, dest, pair_job, pair_read, records, job_side, job_calls, best_full, n, stream=None):
        """Traced records over the score records they replace (pc_phase_b_scatter)."""
        import torch
        assert traced.dtype == torch.int32 and traced.is_contiguous() and records.dtype == torch.int32
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_phase_b_scatter(self._ctx, traced.data_ptr(), int(dest.shape[0]), dest.data_ptr(), pair_job.data_ptr(),
                                          pair_read.data_ptr(), records.data_ptr(), job_side.data_ptr(), job_calls.data_ptr(),
                                          None if best_full is None else best_full.data_ptr(), int(n), ctypes.c_void_p(s)),
              "pc_phase_b_scatter")

    def copy_windows(self, arena, src_off, length, dst, dst_off, pad, stream=None):
        """Packed private copies on the device (pc_copy_windows): window i of `arena` -> dst[dst_off[i]:], padded
        with `pad` up to dst_off[i+1]; src_off int64[n], length int32[n], dst_off int64[n+1], all CUDA tensors."""
        import torch
        n = int(src_off.shape[0])
        assert arena.is_cuda and dst.is_cuda and src_off.dtype == torch.int64 and length.dtype == torch.int32
        assert dst_off.dtype == torch.int64 and int(dst_off.shape[0]) == n + 1
        assert src_off.is_contiguous() and length.is_contiguous() and dst_off.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_copy_windows(self._ctx, arena.data_ptr(), src_off.data_ptr(), length.data_ptr(), n, dst.data_ptr(),
                                       dst_off.data_ptr(), int(pad), ctypes.c_void_p(s)), "pc_copy_windows")

    def unpack_device(self, packed, nbases, exceptions, arena=None, pad=64, stream=None):
        """2 bits per base -> the byte arena the scans take, on the device (pc_unpack_device).  packed: uint8 CUDA tensor
        (io.pack_reads' plane, uploaded); exceptions: int64 CUDA tensor of the non-ACGT positions; arena: optional
        preallocated uint8 CUDA tensor of >= nbases + pad bytes.  -> the arena (nbases bases + `pad` bytes of 'N')."""
        import torch
        assert packed.is_cuda and packed.dtype == torch.uint8 and packed.is_contiguous()
        nbases = int(nbases)
        assert int(packed.numel()) * 4 >= nbases
        if arena is None:
            arena = torch.empty(nbases + pad, dtype=torch.uint8, device=packed.device)
        assert arena.is_cuda and arena.dtype == torch.uint8 and int(arena.numel()) >= nbases + pad
        ne = 0 if exceptions is None else int(exceptions.numel())
        if ne:
            assert exceptions.is_cuda and exceptions.dtype == torch.int64 and exceptions.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_unpack_device(self._ctx, packed.data_ptr(), nbases, exceptions.data_ptr() if ne else None, ne,
                                        arena.data_ptr(), int(pad), ctypes.c_void_p(s)), "pc_unpack_device")
        return arena

    def max_edits(self, adapter_len, threshold_percent):
        """Most non-matching columns an alignment of an adapter_len-base adapter can have inside the adapter's span if
        its full-adapter identity reaches threshold_percent (pc_prefilter_max_edits)."""
        return int(self.lib.pc_prefilter_max_edits(int(adapter_len), float(threshold_percent)))

    def prefilter_mask(self, arena, win_off, win_len, max_len, adapters, max_edits, stream=None):
        """Exact prefilter (pc_prefilter_device) -> int32 CUDA tensor [n, ceil(len(adapters) / 32)]: bit j % 32 of word
        j // 32 of row w is clear when window w is PROVEN not to hold adapters[j] within max_edits[j] edits."""
        import torch
        assert arena.is_cuda and win_off.is_cuda and win_len.is_cuda
        assert win_off.dtype == torch.int64 and win_len.dtype == torch.int32 and win_off.is_contiguous() and win_len.is_contiguous()
        n, na = int(win_off.shape[0]), len(adapters)
        words = (na + 31) // 32
        mask = torch.empty((n, max(words, 1)), dtype=torch.int32, device=arena.device)
        if na == 0:
            return mask.zero_()
        ad = np.ascontiguousarray(adapters, dtype=np.int32)
        ed = np.ascontiguousarray(max_edits, dtype=np.int32)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_prefilter_device(self._ctx, arena.data_ptr(), win_off.data_ptr(), win_len.data_ptr(), n, int(max_len),
                                           ad.ctypes.data, ed.ctypes.data, na, mask.data_ptr(), ctypes.c_void_p(s)),
              "pc_prefilter_device")
        return mask

    def prefilter_mask_packed(self, plane, win_off, win_len, max_len, adapters, max_edits, stream=None):
        """prefilter_mask over reads held at 2 bits per base (pc_prefilter_packed): plane = the uint8 CUDA tensor of
        io.pack_reads' plane (64 readable bytes past the last base), win_off in BASES.  -> the mask, or None when this
        adapter list does not take the packed route (an adapter with a letter other than A/C/G/T/U, or one the seed stage
        cannot cover): unpack and call prefilter_mask."""
        import torch
        assert plane.is_cuda and plane.dtype == torch.uint8 and win_off.is_cuda and win_len.is_cuda
        assert win_off.dtype == torch.int64 and win_len.dtype == torch.int32 and win_off.is_contiguous() and win_len.is_contiguous()
        n, na = int(win_off.shape[0]), len(adapters)
        words = (na + 31) // 32
        mask = torch.empty((n, max(words, 1)), dtype=torch.int32, device=plane.device)
        if na == 0:
            return mask.zero_()
        ad = np.ascontiguousarray(adapters, dtype=np.int32)
        ed = np.ascontiguousarray(max_edits, dtype=np.int32)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        rc = self.lib.pc_prefilter_packed(self._ctx, plane.data_ptr(), win_off.data_ptr(), win_len.data_ptr(), n, int(max_len),
                                          ad.ctypes.data, ed.ctypes.data, na, mask.data_ptr(), ctypes.c_void_p(s))
        if rc == -2:                              # PC_ERR_UNSUPPORTED_SCORES: not a list for the packed route
            return None
        check(rc, "pc_prefilter_packed")
        return mask

    def unpack_windows(self, plane, exceptions, src_off, length, dst, dst_off, pad=ord("N"), stream=None):
        """Windows of the 2-bit plane as bytes (pc_unpack_windows): src_off int64[n] in bases (ascending, not overlapping),
        length int32[n], dst uint8, dst_off int64[n + 1]; 'N' at the listed exceptions (int64, ascending; may be None)."""
        import torch
        n = int(src_off.shape[0])
        assert plane.is_cuda and dst.is_cuda and src_off.dtype == torch.int64 and length.dtype == torch.int32
        assert dst_off.dtype == torch.int64 and int(dst_off.shape[0]) == n + 1
        assert src_off.is_contiguous() and length.is_contiguous() and dst_off.is_contiguous()
        ne = 0 if exceptions is None else int(exceptions.numel())
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_unpack_windows(self._ctx, plane.data_ptr(), exceptions.data_ptr() if ne else None, ne, src_off.data_ptr(),
                                         length.data_ptr(), n, dst.data_ptr(), dst_off.data_ptr(), int(pad), ctypes.c_void_p(s)),
              "pc_unpack_windows")
        return dst

    # ---- the glue of the middle scan as single launches (pc_middle.hip); Pipeline.phase_c keeps the torch formulation for
    # aligners without them (the test stand-ins)
    STAT_BIG = 1 << 40

    def trim_windows(self, off, length, start_trim, end_trim):
        """-> (toff int64[R], tlen int32[R], stats int64[4] on the device: live, longest, STAT_BIG - shortest non-empty, sum)"""
        import torch
        n = int(off.shape[0])
        toff = torch.empty(n, dtype=torch.int64, device=off.device)
        tlen = torch.empty(n, dtype=torch.int32, device=off.device)
        stats = torch.empty(4, dtype=torch.int64, device=off.device)
        off, length = off.contiguous(), length.contiguous()
        st, et = start_trim.to(torch.int32).contiguous(), end_trim.to(torch.int32).contiguous()
        assert off.dtype == torch.int64 and length.dtype == torch.int32
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_trim_windows(self._ctx, off.data_ptr(), length.data_ptr(), st.data_ptr(), et.data_ptr(), n, toff.data_ptr(),
                                       tlen.data_ptr(), stats.data_ptr(), ctypes.c_void_p(s)), "pc_trim_windows")
        return toff, tlen, stats

    def middle_hits(self, rec, threshold):
        """rec int32[..., 8] (contiguous) -> (full float64[...], hit bool[...])"""
        import torch
        assert rec.dtype == torch.int32 and rec.is_contiguous() and rec.shape[-1] == RESULT_INTS
        shape = rec.shape[:-1]
        n = int(rec.numel() // RESULT_INTS)
        full = torch.empty(shape, dtype=torch.float64, device=rec.device)
        hit = torch.empty(shape, dtype=torch.uint8, device=rec.device)
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_middle_hits(self._ctx, rec.data_ptr(), n, float(threshold), full.data_ptr(), hit.data_ptr(), ctypes.c_void_p(s)),
              "pc_middle_hits")
        return full, hit.view(torch.bool)

    def group_survivors(self, mask, gmask):
        """mask int32[n, words], gmask int32[G, words] (device) -> (cand bool[G, n], counts int64[G] on the device)"""
        import torch
        n, words = int(mask.shape[0]), int(mask.shape[1])
        G = int(gmask.shape[0])
        assert mask.dtype == torch.int32 and mask.is_contiguous() and gmask.dtype == torch.int32 and gmask.is_contiguous() and int(gmask.shape[1]) == words
        cand = torch.empty((G, n), dtype=torch.uint8, device=mask.device)
        counts = torch.empty(G, dtype=torch.int64, device=mask.device)
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_group_survivors(self._ctx, mask.data_ptr(), n, words, gmask.data_ptr(), G, cand.data_ptr(), counts.data_ptr(),
                                          ctypes.c_void_p(s)), "pc_group_survivors")
        return cand.view(torch.bool), counts

    def round_consume(self, full_all, rec_all, cur, act, threshold):
        """One consuming step of mask-and-realign (pc_round_consume) -> (anyh bool[n], a_hit int32[n], cnt int64[n], stats int64[4])"""
        import torch
        A, Dn = int(full_all.shape[0]), int(full_all.shape[1])
        n = int(act.shape[0])
        assert full_all.dtype == torch.float64 and full_all.is_contiguous() and rec_all.dtype == torch.int32 and rec_all.is_contiguous()
        assert cur.dtype == torch.int64 and cur.is_contiguous() and act.dtype == torch.int64
        act = act.contiguous()
        anyh = torch.empty(n, dtype=torch.uint8, device=act.device)
        a_hit = torch.empty(n, dtype=torch.int32, device=act.device)
        cnt = torch.empty(n, dtype=torch.int64, device=act.device)
        stats = torch.empty(4, dtype=torch.int64, device=act.device)
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_round_consume(self._ctx, full_all.data_ptr(), rec_all.data_ptr(), cur.data_ptr(), act.data_ptr(), n, A, Dn, float(threshold),
                                        anyh.data_ptr(), a_hit.data_ptr(), cnt.data_ptr(), stats.data_ptr(), ctypes.c_void_p(s)), "pc_round_consume")
        return anyh.view(torch.bool), a_hit, cnt, stats

    def prefilter_defer_count(self, enabled=True):
        """No host round trip inside prefilter_mask (pc_prefilter_defer_count): after the caller's next synchronisation,
        prefilter_overflowed() says whether the last mask is incomplete (then: call again with the deferral off)."""
        check(self.lib.pc_prefilter_defer_count(self._ctx, 1 if enabled else 0), "pc_prefilter_defer_count")

    def prefilter_overflowed(self):
        return bool(self.lib.pc_prefilter_overflowed(self._ctx))

    def prefilter_rows(self, arena, win_off, win_len, max_len, adapters, max_edits, stream=None, packed=False):
        """The prefilter's survivors, sparsely: -> (rows int64 [R]: the windows with at least one surviving adapter, in
        increasing order; bits bool [R, len(adapters)]: which).  Everything not listed is PROVEN not to be a hit."""
        import torch
        if packed:              # arena is the 2-bit plane, win_off counts bases; None = this list does not take the packed route
            mask = self.prefilter_mask_packed(arena, win_off, win_len, max_len, adapters, max_edits, stream)
            if mask is None:
                return None
        else:
            mask = self.prefilter_mask(arena, win_off, win_len, max_len, adapters, max_edits, stream)
        na = len(adapters)
        rows = torch.nonzero((mask != 0).any(dim=1)).flatten()
        sub = mask[rows]
        shifts = torch.arange(32, device=mask.device, dtype=torch.int32)
        bits = ((sub[:, :, None] >> shifts[None, None, :]) & 1).reshape(int(rows.shape[0]), 32 * int(mask.shape[1]))[:, :na].to(torch.bool)
        return rows, bits

    def prefilter(self, arena, win_off, win_len, max_len, adapters, max_edits, stream=None):
        """Dense form of prefilter_rows: bool CUDA tensor [len(adapters), n] (small batches, tests)."""
        import torch
        rows, bits = self.prefilter_rows(arena, win_off, win_len, max_len, adapters, max_edits, stream)
        out = torch.zeros((len(adapters), int(win_off.shape[0])), dtype=torch.bool, device=arena.device)
        out[:, rows] = bits.t()
        return out

    def debug_value_range(self):
        """(lo, hi) of the DP values the range-checking kernel builds have held since the last call."""
        lo, hi = ctypes.c_int32(), ctypes.c_int32()
        check(self.lib.pc_debug_value_range(self._ctx, ctypes.byref(lo), ctypes.byref(hi)), "pc_debug_value_range")
        return lo.value, hi.value

    def trace_ops_per_2_cells(self):
        """Packed VALU ops the traced end-window kernel spends per two DP cells (roofline reporting)."""
        return self.lib.pc_trace_ops_x100(self._ctx) / 100.0

    def sync(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.pc_sync(self._ctx, ctypes.c_void_p(s)), "pc_sync")
