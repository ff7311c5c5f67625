"""GPU-backed stand-in for the one function of the reference's ctypes wrapper that sits on the hot
path.  Interface kept exactly (porechop/cpp_function_wrappers.py:42-53):

    adapter_alignment(read_sequence: str, adapter_sequence: str,
                      scoring_scheme_vals: [match, mismatch, gap_open, gap_extend]) -> str

returning 'readStart,readEnd,adapterStart,adapterEnd,rawScore,alignedRegion%id,fullAdapter%id'.
``porechop.nanopore_read`` imports this name from its sibling module (nanopore_read.py:17);
INTEGRATION.md shows how to put this library underneath (swap the .so, or
``porechop_amd.dropin.install()`` which also batches the phase scans).
"""
import ctypes

from ._lib import load_library


def adapter_alignment(read_sequence, adapter_sequence, scoring_scheme_vals):
    lib = load_library()
    m, x, go, ge = (int(v) for v in scoring_scheme_vals[:4])
    raw = lib.adapterAlignment(read_sequence.encode("utf-8"), adapter_sequence.encode("utf-8"), m, x, go, ge)
    if not raw:
        # the C side already explained why on stderr; there is nothing to fall back to
        raise RuntimeError("porechop_amd: adapterAlignment failed for scoring scheme %r "
                           "(unsupported scheme or no usable GPU)" % (list(scoring_scheme_vals),))
    try:
        return ctypes.string_at(raw).decode()
    finally:
        lib.freeCString(raw)      # the callee malloc()s, the caller frees exactly once
