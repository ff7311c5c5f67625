"""porechop_amd -- MI355X-native adapter-alignment core for Porechop.

Only the one hot path of rrwick/Porechop lives here: ``cpp_function_wrappers.adapter_alignment``
(porechop/cpp_function_wrappers.py:42-53) and what is behind it, rebuilt as HIP kernels for
gfx950 behind a C ABI (include/porechop_amd.h).  There is no CPU fallback: importing the
compute entry points without the built library, or calling them without a GPU, fails loudly.
"""
from ._lib import LIB_PATH, load_library, LibraryMissing  # noqa: F401
from .cpp_function_wrappers import adapter_alignment  # noqa: F401
from .batch import (Aligner, RESULT_INTS, MODE_AUTO, MODE_TRACE, MODE_TWO_PASS, MODE_SCORE,  # noqa: F401
                    format_result, records_to_fields)

__all__ = ["adapter_alignment", "Aligner", "load_library", "LIB_PATH", "format_result"]
