"""Mirror of the reference's porechop/cpp_function_wrappers.py for the one function on the hot
path: same name, same arguments, same 7-field string (porechop/cpp_function_wrappers.py:42-63),
but the C symbol behind it is this repository's GPU library.

``porechop.nanopore_read`` imports ``adapter_alignment`` from its sibling module
(porechop/nanopore_read.py:17); INTEGRATION.md shows the two ways to drop this in (swap the
.so, or ``porechop_amd.dropin.install()`` which also batches the phase scans).
"""
from ctypes import c_char_p, cast

from ._lib import load_library


def adapter_alignment(read_sequence, adapter_sequence, scoring_scheme_vals):
    """Python wrapper for the adapterAlignment C function (GPU-backed).

    scoring_scheme_vals = [match, mismatch, gap_open, gap_extend]; returns
    'readStart,readEnd,adapterStart,adapterEnd,rawScore,alignedRegion%id,fullAdapter%id'.
    """
    lib = load_library()
    match_score = scoring_scheme_vals[0]
    mismatch_score = scoring_scheme_vals[1]
    gap_open_score = scoring_scheme_vals[2]
    gap_extend_score = scoring_scheme_vals[3]
    ptr = lib.adapterAlignment(read_sequence.encode('utf-8'), adapter_sequence.encode('utf-8'),
                               match_score, mismatch_score, gap_open_score, gap_extend_score)
    if not ptr:
        raise RuntimeError('porechop_amd: adapterAlignment failed (unsupported scoring scheme %r '
                           'or no usable GPU); there is no CPU fallback' % (scoring_scheme_vals,))
    return c_string_to_python_string(ptr)


def c_string_to_python_string(c_string):
    """Decode the returned C string, then free it (as the reference wrapper does)."""
    lib = load_library()
    python_string = cast(c_string, c_char_p).value.decode()
    lib.freeCString(c_string)
    return python_string
