"""CPU-side checks of the boundary: the built library loads and exports every symbol that
include/porechop_amd.h declares; without a GPU every compute entry point fails loudly instead
of falling back to anything."""
import ctypes
import os
import re

import pytest

import porechop_amd
from porechop_amd._lib import EXPORTS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "porechop_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = "\n".join(l for l in hdr.split("\n") if not l.lstrip().startswith("#"))
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", hdr)
    return sorted(set(names))


def test_header_and_binding_list_agree():
    assert declared_symbols() == sorted(EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = porechop_amd.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pc_version()


def test_dropin_twin_exists_and_exports_reference_symbols():
    # porechop/cpp_function_wrappers.py:21-33 loads "cpp_functions.so" and binds these two
    twin = os.path.join(os.path.dirname(porechop_amd.LIB_PATH), "cpp_functions.so")
    assert os.path.isfile(twin)
    L = ctypes.CDLL(twin)
    assert hasattr(L, "adapterAlignment") and hasattr(L, "freeCString")


def test_score_scheme_gate():
    lib = porechop_amd.load_library()
    assert lib.pc_scores_supported(3, -6, -5, -2, 111) == 1      # Porechop default
    assert lib.pc_scores_supported(3, -6, -5, -5, 28) == 1       # linear gaps: the reference's NW dispatch
    assert lib.pc_scores_supported(3, -6, 5, -2, 28) == 0        # non-negative gap score
    assert lib.pc_scores_supported(3, 3, -5, -2, 28) == 0        # match <= mismatch
    assert lib.pc_scores_supported(300, -6, -5, -2, 111) == 0    # would overflow int16 lanes
    assert lib.pc_scores_supported(3, -6, -5, -2, 500) == 0      # adapter too long


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is exercised on the CPU-only box")
    with pytest.raises(RuntimeError):
        porechop_amd.Aligner(["ACGT"])
    with pytest.raises(RuntimeError):
        porechop_amd.adapter_alignment("ACGT", "ACGT", [3, -6, -5, -2])


def test_context_functions_refuse_a_null_context():
    """Every batch entry point checks its context before it touches the device (no compute without a GPU here): the
    round's new ones -- the exact prefilter and the three kernels of phase B's exact pruning -- included."""
    lib = porechop_amd.load_library()
    BAD = -3                                                     # PC_ERR_BAD_ARG (include/porechop_amd.h)
    assert lib.pc_phase_b_select(None, None, 0, 0, None, None, None, None, None, None, 150, 4, 2, 75.0, 1, 1e9, 0.0,
                                 None, None, None, None, None, None, None, None, None) == BAD
    assert lib.pc_phase_b_gather(None, None, 0, 0, *([None] * 14)) == BAD
    assert lib.pc_phase_b_scatter(None, None, 0, None, None, None, None, None, None, None, 0, None) == BAD
    assert lib.pc_phase_b_reduce(None, None, 0, 0, None, None, 150, 4, 2, 75.0, None, None, 0, None, None, 0.0, 0.0, 0, None, None) == BAD
    assert lib.pc_prefilter_device(None, None, None, None, 0, 0, None, None, 0, None, None) == BAD
    assert lib.pc_strerror(BAD).decode() == "bad argument"
