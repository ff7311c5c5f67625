"""Host ingest: FASTA/FASTQ(.gz) -> packed arena + offset/length tables (the C side is
porechop_amd/csrc/pc_io.cpp; semantics of porechop/misc.py:60-168 + nanopore_read.py:23-35)."""
import ctypes

import numpy as np

from ._lib import load_library


class ReadSet:
    """Reads of one file, normalised like NanoporeRead.__init__, as numpy views over one arena."""

    def __init__(self, path):
        self.lib = load_library()
        self._h = ctypes.c_void_p()
        rc = self.lib.pc_readset_load(str(path).encode(), ctypes.byref(self._h))
        if rc != 0:
            msg = self.lib.pc_readset_error(self._h).decode() if self._h else "load failed"
            self.close()
            raise ValueError("Error: " + msg)
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
        else:
            self.offsets = np.zeros(0, dtype=np.int64)
            self.lengths = np.zeros(0, dtype=np.int32)

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

    def to_device(self, device="cuda"):
        """-> porechop_amd.pipeline.DeviceReads (one upload of the packed arena)."""
        import torch
        from .pipeline import DeviceReads
        dev = torch.device(device)
        return DeviceReads(torch.from_numpy(self.arena.copy()).to(dev), torch.from_numpy(self.offsets.copy()).to(dev),
                           torch.from_numpy(self.lengths.copy()).to(dev))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pc_readset_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
