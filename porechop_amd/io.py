"""Host ingest: FASTA/FASTQ(.gz) -> packed arena + offset/length tables, and the output writer
(the C side is porechop_amd/csrc/pc_io.cpp; semantics of porechop/misc.py:60-168,
nanopore_read.py:23-35 and, for writing, nanopore_read.py:97-147 + porechop.py:607-734)."""
import ctypes
import os

import numpy as np

from ._lib import load_library


def pack_reads(arena, nbases=None, out=None):
    """One byte per base -> 2 bits per base + the positions of the non-ACGT(U) bases (pc_pack_reads; all host cores).
    arena: uint8 numpy array (or bytes); nbases: how many of its bytes are bases (default: all).
    -> (packed uint8 [ceil(nbases / 16) * 4], exceptions int64 [E], ascending).  `out` may be a preallocated uint8 array
    (pinned memory, say) of at least that many bytes."""
    lib = load_library()
    a = np.frombuffer(arena, dtype=np.uint8) if isinstance(arena, (bytes, bytearray)) else np.ascontiguousarray(arena, dtype=np.uint8)
    n = int(a.size if nbases is None else nbases)
    assert 0 <= n <= a.size
    nb = (n + 15) // 16 * 4
    packed = np.zeros(nb, dtype=np.uint8) if out is None else out
    assert packed.dtype == np.uint8 and packed.size >= nb and packed.flags["C_CONTIGUOUS"]
    nexc = ctypes.c_int64()
    exc = np.zeros(max(16, n // 4096), dtype=np.int64)
    rc = lib.pc_pack_reads(a.ctypes.data, n, packed.ctypes.data, exc.ctypes.data, exc.size, ctypes.byref(nexc))
    if rc != 0 and nexc.value > exc.size:
        exc = np.zeros(nexc.value, dtype=np.int64)
        rc = lib.pc_pack_reads(a.ctypes.data, n, packed.ctypes.data, exc.ctypes.data, exc.size, ctypes.byref(nexc))
    if rc != 0:
        raise RuntimeError("pc_pack_reads failed (%d)" % rc)
    return packed[:nb], exc[:nexc.value].copy()


def unpack_reads_host(packed, nbases, exc):
    """The inverse of pack_reads in numpy (what pc_unpack_device writes, without the padding): for tests and hosts
    that want the canonical bytes back."""
    p = np.ascontiguousarray(packed, dtype=np.uint8)
    codes = ((p[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)[None, :]) & 3).reshape(-1)[:nbases]
    out = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].copy()
    out[np.asarray(exc, dtype=np.int64)] = ord("N")
    return out


def fastq_record_start(path, byte_pos):
    """First record start at or after byte_pos of a plain 4-line FASTQ file (the file size when none is left), or None when
    the file is not streamable there (gzip, FASTA, an irregular record): pc_fastq_find_record."""
    lib = load_library()
    out = ctypes.c_int64()
    rc = lib.pc_fastq_find_record(str(path).encode(), int(byte_pos), ctypes.byref(out))
    return int(out.value) if rc == 0 else None


def gz_member_start(path, pos):
    """First gzip member that starts at or after COMPRESSED byte pos (validated by inflating it; the file's size when there is
    none), or None when the file is not a gzip file: pc_gz_member_start."""
    lib = load_library()
    out = ctypes.c_int64()
    rc = lib.pc_gz_member_start(str(path).encode(), int(pos), ctypes.byref(out))
    return int(out.value) if rc == 0 else None


def gz_sized_size(path):
    """Inflated size of a gzip file made of sized members only (this library's output; bgzip), else None: pc_gz_sized_size."""
    lib = load_library()
    out = ctypes.c_int64()
    return int(out.value) if lib.pc_gz_sized_size(str(path).encode(), ctypes.byref(out)) == 0 else None


def gz_sized_record_start(path, pos):
    """fastq_record_start on the INFLATED bytes of such a file (only the members around pos are inflated), or None."""
    lib = load_library()
    out = ctypes.c_int64()
    return int(out.value) if lib.pc_gz_sized_find_record(str(path).encode(), int(pos), ctypes.byref(out)) == 0 else None


GZ_LEVEL = int(os.environ.get("PC_GZ_LEVEL", "0"))      # 0: the library's default (pc_gz.h default_level)


def gzip_file(src, dst, level=None, single_member=False):
    """src -> dst through the library's parallel compressor (pc_gzip_file): sized members, or ONE pigz-style member."""
    rc = load_library().pc_gzip_file(str(src).encode(), str(dst).encode(), int(level if level is not None else GZ_LEVEL), 1 if single_member else 0)
    if rc != 0:
        raise OSError("Error: could not compress " + str(src))


def gz_finish(path):
    """The empty last member of a file of sized members (pc_gz_finish); creates the file when there is none."""
    if load_library().pc_gz_finish(str(path).encode()) != 0:
        raise OSError("Error: could not write " + str(path))


class GzImage:
    """The compressed image of some pieces, per file, in memory (pc_readset_compress)."""

    def __init__(self, lib, handle, nfiles):
        self.lib, self._h, self.nfiles = lib, handle, nfiles

    def sizes(self):
        """-> (compressed bytes, plain bytes) per file, numpy int64 [nfiles]."""
        z = np.zeros(max(1, self.nfiles), dtype=np.int64)
        pl = np.zeros(max(1, self.nfiles), dtype=np.int64)
        if self.lib.pc_gzimage_sizes(self._h, self.nfiles, z.ctypes.data, pl.ctypes.data) != 0:
            raise OSError("Error: could not size the compressed output")
        return z[:self.nfiles], pl[:self.nfiles]

    def write(self, file_paths, file_pos, shared=False):
        """Every file's image at file_pos[f] (numpy int64, updated in place); shared: other processes write other spans."""
        assert file_pos.dtype == np.int64 and file_pos.flags["C_CONTIGUOUS"] and file_pos.shape[0] == len(file_paths) == self.nfiles
        paths = (ctypes.c_char_p * max(1, len(file_paths)))(*[str(p).encode() for p in file_paths])
        if self.lib.pc_gzimage_write(self._h, self.nfiles, paths, file_pos.ctypes.data, 1 if shared else 0) != 0:
            raise OSError("Error: could not write the output reads")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pc_gzimage_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GzStream:
    """A gzip FASTQ file as a stream of ReadSet blocks (pc_gzstream_*): inflated ahead of the caller by a producer thread."""

    def __init__(self, path, begin=None, end=None):
        """begin / end: only the members in [begin, end) of the COMPRESSED bytes (member starts, gz_member_start; end None =
        to the end of the file): one rank's share of a multi-member gzip file without sizes."""
        self.lib = load_library()
        self.path = str(path)
        self._h = ctypes.c_void_p()
        if begin is None:
            rc = self.lib.pc_gzstream_open(self.path.encode(), ctypes.byref(self._h))
        else:
            rc = self.lib.pc_gzstream_open_range(self.path.encode(), int(begin), int(end) if end is not None else 0, ctypes.byref(self._h))
        if rc != 0:
            self._h = None
            raise ValueError("not a gzip file: " + self.path)

    def next(self, target_bytes, min_reads=0):
        """-> a ReadSet; None at the end of the file; False when the input is not streamable (irregular FASTQ, damaged stream)."""
        h = ctypes.c_void_p()
        eof = ctypes.c_int()
        rc = self.lib.pc_gzstream_next(self._h, int(target_bytes), int(min_reads), ctypes.byref(h), ctypes.byref(eof))
        if rc != 0:
            return False
        if eof.value or not h:
            return None
        return ReadSet(self.path, _handle=h)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pc_gzstream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadSet:
    """Reads of one file -- or of several files, in order (Albacore directory input) -- normalised
    like NanoporeRead.__init__, as numpy views over one arena."""

    def __init__(self, path, _handle=None):
        self.lib = load_library()
        self._h = ctypes.c_void_p()
        paths = [path] if isinstance(path, (str, bytes)) or hasattr(path, "__fspath__") else list(path)
        self.paths = [str(p) for p in paths]
        if _handle is not None:
            self._h = _handle
        else:
            arr = (ctypes.c_char_p * len(self.paths))(*[p.encode() for p in self.paths])
            rc = self.lib.pc_readset_load_many(arr, len(self.paths), ctypes.byref(self._h))
            if rc != 0:
                msg = self.lib.pc_readset_error(self._h).decode() if self._h else "load failed"
                self.close()
                raise ValueError("Error: " + msg)
        self._bind()

    @classmethod
    def segment(cls, path, byte_begin, target_bytes):
        """One block of a plain 4-line FASTQ file (pc_readset_load_segment): the records starting in
        [byte_begin, about byte_begin + target_bytes) -> (ReadSet or None if the file is not streamable, next_begin)."""
        lib = load_library()
        h = ctypes.c_void_p()
        nxt = ctypes.c_int64()
        rc = lib.pc_readset_load_segment(str(path).encode(), int(byte_begin), int(target_bytes), ctypes.byref(nxt), ctypes.byref(h))
        if rc != 0:
            if h:
                lib.pc_readset_free(h)
            return None, int(byte_begin)
        return cls(path, _handle=h), int(nxt.value)

    @classmethod
    def gz_range(cls, path, begin, end):
        """The records of [begin, end) of the inflated bytes of a gzip file of sized members (both from
        gz_sized_record_start): pc_readset_load_gz_range -> ReadSet, or None."""
        lib = load_library()
        h = ctypes.c_void_p()
        rc = lib.pc_readset_load_gz_range(str(path).encode(), int(begin), int(end), ctypes.byref(h))
        if rc != 0:
            if h:
                lib.pc_readset_free(h)
            return None
        return cls(path, _handle=h)

    def _bind(self):
        n = self.lib.pc_readset_count(self._h)
        self.count = int(n)
        self.is_fastq = bool(self.lib.pc_readset_is_fastq(self._h))
        nbytes = ctypes.c_int64()
        ap = self.lib.pc_readset_arena(self._h, ctypes.byref(nbytes))
        self.arena = np.ctypeslib.as_array(ctypes.cast(ap, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes.value,))
        if n:
            self.offsets = np.ctypeslib.as_array(
                ctypes.cast(self.lib.pc_readset_offsets(self._h), ctypes.POINTER(ctypes.c_int64)), shape=(n,))
            self.lengths = np.ctypeslib.as_array(
                ctypes.cast(self.lib.pc_readset_lengths(self._h), ctypes.POINTER(ctypes.c_int32)), shape=(n,))
            self.file_index = np.ctypeslib.as_array(
                ctypes.cast(self.lib.pc_readset_file_index(self._h), ctypes.POINTER(ctypes.c_int32)), shape=(n,))
        else:
            self.offsets = np.zeros(0, dtype=np.int64)
            self.lengths = np.zeros(0, dtype=np.int32)
            self.file_index = np.zeros(0, dtype=np.int32)

    def name(self, i):
        return self.lib.pc_readset_name(self._h, i).decode()

    def seq(self, i):
        o, n = int(self.offsets[i]), int(self.lengths[i])
        return self.arena[o:o + n].tobytes().decode()

    def quals(self, i):
        q = self.lib.pc_readset_quals(self._h, i)
        return q.decode() if q is not None else None

    def is_rna(self, i):
        return bool(self.lib.pc_readset_is_rna(self._h, i))

    def to_device(self, device="cuda", packed=False, aligner=None):
        """-> porechop_amd.pipeline.DeviceReads.  packed=True sends the bases over PCIe at 2 bits each (pack_reads on the
        host, pc_unpack_device on the GPU; `aligner` = any porechop_amd.Aligner of that device) -- a quarter of the bytes,
        alignment-equivalent contents (non-ACGT bases arrive as 'N')."""
        import torch
        from .pipeline import DeviceReads
        dev = torch.device(device)
        off = torch.from_numpy(self.offsets.copy()).to(dev)
        ln = torch.from_numpy(self.lengths.copy()).to(dev)
        if not packed:
            return DeviceReads(torch.from_numpy(self.arena.copy()).to(dev), off, ln)
        nbases = int(self.offsets[-1] + self.lengths[-1]) if self.count else 0
        pk, exc = pack_reads(self.arena, nbases)
        return DeviceReads.from_packed(aligner, torch.from_numpy(pk).to(dev), nbases, torch.from_numpy(exc).to(dev), off, ln)

    def write(self, piece_read, piece_start, piece_len, piece_number, piece_file, file_paths, fastq):
        """Write pieces of reads (see pc_readset_write in include/porechop_amd.h) -> bytes written."""
        pr = np.ascontiguousarray(piece_read, dtype=np.int64)
        ps = np.ascontiguousarray(piece_start, dtype=np.int32)
        pl = np.ascontiguousarray(piece_len, dtype=np.int32)
        pn = np.ascontiguousarray(piece_number, dtype=np.int32)
        pf = np.ascontiguousarray(piece_file, dtype=np.int32)
        assert pr.shape == ps.shape == pl.shape == pn.shape == pf.shape
        paths = (ctypes.c_char_p * max(1, len(file_paths)))(*[str(p).encode() for p in file_paths])
        written = ctypes.c_int64()
        rc = self.lib.pc_readset_write(self._h, pr.shape[0], pr.ctypes.data, ps.ctypes.data, pl.ctypes.data,
                                       pn.ctypes.data, pf.ctypes.data, len(file_paths), paths, 1 if fastq else 0,
                                       ctypes.byref(written))
        if rc != 0:
            raise OSError("Error: could not write the output reads")
        return written.value

    def write_at(self, piece_read, piece_start, piece_len, piece_number, piece_file, file_paths, fastq, file_pos):
        """write() for a streamed run: file_pos (numpy int64 [len(file_paths)]) says where each file continues
        (0 = create it) and is updated in place (pc_readset_write_at)."""
        pr = np.ascontiguousarray(piece_read, dtype=np.int64)
        ps = np.ascontiguousarray(piece_start, dtype=np.int32)
        pl = np.ascontiguousarray(piece_len, dtype=np.int32)
        pn = np.ascontiguousarray(piece_number, dtype=np.int32)
        pf = np.ascontiguousarray(piece_file, dtype=np.int32)
        assert pr.shape == ps.shape == pl.shape == pn.shape == pf.shape
        assert file_pos.dtype == np.int64 and file_pos.flags["C_CONTIGUOUS"] and file_pos.shape[0] == len(file_paths)
        paths = (ctypes.c_char_p * max(1, len(file_paths)))(*[str(p).encode() for p in file_paths])
        rc = self.lib.pc_readset_write_at(self._h, pr.shape[0], pr.ctypes.data, ps.ctypes.data, pl.ctypes.data,
                                          pn.ctypes.data, pf.ctypes.data, len(file_paths), paths, 1 if fastq else 0,
                                          file_pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        if rc != 0:
            raise OSError("Error: could not write the output reads")

    def write_sizes(self, piece_read, piece_start, piece_len, piece_number, piece_file, nfiles, fastq):
        """Bytes write() would put into each of nfiles files (numpy int64 [nfiles]); nothing is written (pc_readset_write_sizes)."""
        pr = np.ascontiguousarray(piece_read, dtype=np.int64)
        ps = np.ascontiguousarray(piece_start, dtype=np.int32)
        pl = np.ascontiguousarray(piece_len, dtype=np.int32)
        pn = np.ascontiguousarray(piece_number, dtype=np.int32)
        pf = np.ascontiguousarray(piece_file, dtype=np.int32)
        out = np.zeros(max(1, nfiles), dtype=np.int64)
        rc = self.lib.pc_readset_write_sizes(self._h, pr.shape[0], pr.ctypes.data, ps.ctypes.data, pl.ctypes.data, pn.ctypes.data,
                                             pf.ctypes.data, int(nfiles), 1 if fastq else 0, out.ctypes.data)
        if rc != 0:
            raise OSError("Error: could not size the output reads")
        return out[:nfiles]

    def compress(self, piece_read, piece_start, piece_len, piece_number, piece_file, nfiles, fastq, level=None):
        """The pieces write() would write, formatted and deflated by all cores into memory -> GzImage (pc_readset_compress)."""
        pr = np.ascontiguousarray(piece_read, dtype=np.int64)
        ps = np.ascontiguousarray(piece_start, dtype=np.int32)
        pl = np.ascontiguousarray(piece_len, dtype=np.int32)
        pn = np.ascontiguousarray(piece_number, dtype=np.int32)
        pf = np.ascontiguousarray(piece_file, dtype=np.int32)
        assert pr.shape == ps.shape == pl.shape == pn.shape == pf.shape
        h = ctypes.c_void_p()
        rc = self.lib.pc_readset_compress(self._h, pr.shape[0], pr.ctypes.data, ps.ctypes.data, pl.ctypes.data, pn.ctypes.data,
                                          pf.ctypes.data, int(nfiles), 1 if fastq else 0, int(level if level is not None else GZ_LEVEL), ctypes.byref(h))
        img = GzImage(self.lib, h, int(nfiles))
        if rc != 0:
            img.close()
            raise OSError("Error: could not compress the output reads")
        return img

    def write_shared(self, piece_read, piece_start, piece_len, piece_number, piece_file, file_paths, fastq, file_pos):
        """write_at() into files that other processes write disjoint spans of (a sharded run): never truncates
        (pc_readset_write_shared)."""
        pr = np.ascontiguousarray(piece_read, dtype=np.int64)
        ps = np.ascontiguousarray(piece_start, dtype=np.int32)
        pl = np.ascontiguousarray(piece_len, dtype=np.int32)
        pn = np.ascontiguousarray(piece_number, dtype=np.int32)
        pf = np.ascontiguousarray(piece_file, dtype=np.int32)
        assert file_pos.dtype == np.int64 and file_pos.flags["C_CONTIGUOUS"] and file_pos.shape[0] == len(file_paths)
        paths = (ctypes.c_char_p * max(1, len(file_paths)))(*[str(p).encode() for p in file_paths])
        rc = self.lib.pc_readset_write_shared(self._h, pr.shape[0], pr.ctypes.data, ps.ctypes.data, pl.ctypes.data,
                                              pn.ctypes.data, pf.ctypes.data, len(file_paths), paths, 1 if fastq else 0,
                                              file_pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
        if rc != 0:
            raise OSError("Error: could not write the output reads")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pc_readset_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
