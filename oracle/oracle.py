"""ctypes bindings for the parity checkers under oracle/  (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under porechop_amd/ does.

  * ``Oracle``     -> oracle/_build/libpc_oracle.so, our own C restatement (pc_oracle.c)
  * ``Reference``  -> oracle/_ref/cpp_functions.so, the reference's own sources compiled by
                      oracle/Makefile ``make ref`` (same C ABI the reference wrapper binds at
                      porechop/cpp_function_wrappers.py:27-39)
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_build", "libpc_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "cpp_functions.so")
REFERENCE_ROOT = "/root/reference"

DEFAULT_SCORES = (3, -6, -5, -2)  # match, mismatch, gap_open, gap_extend (porechop.py:145)


def build_oracle(force=False):
    if force or not os.path.isfile(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "pc_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return ORACLE_SO


def build_ref():
    """Compile the reference's own sources into oracle/_ref (only possible where
    /root/reference exists, i.e. in the build container; the GPU box uses the prebuilt file)."""
    if os.path.isfile(REF_SO):
        return REF_SO
    if not os.path.isdir(REFERENCE_ROOT):
        return None
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    return REF_SO


class _Result(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "read_start", "read_end", "adapter_start", "adapter_end", "score",
        "aligned_matches", "aligned_len", "full_matches", "full_len", "path_len", "failed", "end_i", "end_j")]


class Oracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.pc_oracle_adapterAlignment.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 4
        L.pc_oracle_adapterAlignment.restype = ctypes.c_void_p
        L.pc_oracle_free.argtypes = [ctypes.c_void_p]
        L.pc_oracle_align_raw.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int] + \
            [ctypes.c_int] * 4 + [ctypes.POINTER(_Result)]
        L.pc_oracle_align_raw.restype = ctypes.c_int
        L.pc_oracle_align_many.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] + [ctypes.c_int] * 4 + \
            [ctypes.c_void_p]
        L.pc_oracle_align_many.restype = ctypes.c_int
        L.pc_oracle_min_edits.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.pc_oracle_min_edits.restype = ctypes.c_int
        L.pc_oracle_min_edits_many.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
        L.pc_oracle_min_edits_many.restype = ctypes.c_int

    def adapter_alignment(self, read, adapter, scores=DEFAULT_SCORES):
        if isinstance(read, str):
            read = read.encode()
        if isinstance(adapter, str):
            adapter = adapter.encode()
        p = self.lib.pc_oracle_adapterAlignment(read, adapter, *scores)
        if not p:
            raise RuntimeError("oracle: unsupported scoring scheme %r" % (scores,))
        s = ctypes.cast(p, ctypes.c_char_p).value.decode()
        self.lib.pc_oracle_free(p)
        return s

    def align_raw(self, read, adapter, scores=DEFAULT_SCORES):
        if isinstance(read, str):
            read = read.encode()
        if isinstance(adapter, str):
            adapter = adapter.encode()
        r = _Result()
        rc = self.lib.pc_oracle_align_raw(read, len(read), adapter, len(adapter), *scores, ctypes.byref(r))
        if rc:
            raise RuntimeError("oracle rc=%d" % rc)
        return r

    def min_edits(self, read, adapter):
        """Smallest unit-cost edit distance between the whole adapter and any substring of the read (the quantity
        the exact prefilter bounds; pc_oracle.c)."""
        if isinstance(read, str):
            read = read.encode()
        if isinstance(adapter, str):
            adapter = adapter.encode()
        return int(self.lib.pc_oracle_min_edits(read, len(read), adapter, len(adapter)))

    def min_edits_many(self, read_arena, read_off, read_len, adapter):
        read_arena = np.ascontiguousarray(read_arena, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        if isinstance(adapter, str):
            adapter = adapter.encode()
        out = np.zeros(read_off.shape[0], dtype=np.int32)
        rc = self.lib.pc_oracle_min_edits_many(read_arena.ctypes.data, read_off.ctypes.data, read_len.ctypes.data,
                                               adapter, len(adapter), read_off.shape[0], out.ctypes.data)
        if rc:
            raise RuntimeError("oracle rc=%d" % rc)
        return out

    def align_many(self, read_arena, read_off, read_len, ad_arena, ad_off, ad_len, scores=DEFAULT_SCORES):
        """Bulk API over numpy arrays; returns int32 [npairs, 9]
        (rs, re, as, ae, score, aligned_matches, aligned_len, full_matches, full_len)."""
        read_arena = np.ascontiguousarray(read_arena, dtype=np.uint8)
        ad_arena = np.ascontiguousarray(ad_arena, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        read_len = np.ascontiguousarray(read_len, dtype=np.int32)
        ad_off = np.ascontiguousarray(ad_off, dtype=np.int64)
        ad_len = np.ascontiguousarray(ad_len, dtype=np.int32)
        n = read_off.shape[0]
        out = np.zeros((n, 9), dtype=np.int32)
        rc = self.lib.pc_oracle_align_many(
            read_arena.ctypes.data, read_off.ctypes.data, read_len.ctypes.data,
            ad_arena.ctypes.data, ad_off.ctypes.data, ad_len.ctypes.data,
            n, *scores, out.ctypes.data)
        if rc:
            raise RuntimeError("oracle rc=%d" % rc)
        return out


class Reference:
    """The compiled reference (None-safe: ``Reference.available()``)."""

    @staticmethod
    def available():
        return build_ref() is not None

    def __init__(self):
        so = build_ref()
        if so is None:
            raise RuntimeError("reference .so unavailable (no oracle/_ref and no /root/reference)")
        self.lib = ctypes.CDLL(so)
        self.lib.adapterAlignment.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_int] * 4
        self.lib.adapterAlignment.restype = ctypes.c_void_p
        self.lib.freeCString.argtypes = [ctypes.c_void_p]
        self.lib.freeCString.restype = None

    def adapter_alignment(self, read, adapter, scores=DEFAULT_SCORES):
        if isinstance(read, str):
            read = read.encode()
        if isinstance(adapter, str):
            adapter = adapter.encode()
        p = self.lib.adapterAlignment(read, adapter, *scores)
        s = ctypes.cast(p, ctypes.c_char_p).value.decode()
        self.lib.freeCString(p)
        return s


def parse_fields(s):
    """Split a 7-field result into comparable pieces: ints stay ints, identities stay the
    exact decimal strings the C side printed."""
    f = s.split(",")
    return (int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], f[6])
