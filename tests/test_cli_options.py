"""`python -m porechop_amd` takes the reference's command line: every option of porechop/porechop.py:82-203 with the same name,
destination, type, default, choices and arity -- introspected from both argparse parsers (only where /root/reference exists).
The one intended difference: --threads defaults to 1 (accepted for compatibility; the alignments run on the GPU)."""
import argparse
import os
import sys
import tempfile

import pytest

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")


def _parsers():
    captured = {}
    orig = argparse.ArgumentParser.parse_args

    def grab(self, args=None, namespace=None):
        captured[captured["which"]] = self
        raise SystemExit(0)
    argparse.ArgumentParser.parse_args = grab
    old_argv, old_path = sys.argv, list(sys.path)
    try:
        from tests.golden.make_golden import stage_reference
        ref = stage_reference(tempfile.mkdtemp(prefix="pc_cli_"))
        sys.path.insert(0, ref)
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
        import porechop.porechop as pp
        captured["which"] = "ref"
        sys.argv = ["porechop", "-i", "x"]
        try:
            pp.get_arguments()
        except SystemExit:
            pass
        import porechop_amd.__main__ as ours
        captured["which"] = "ours"
        try:
            ours.main(["-i", "x"])
        except SystemExit:
            pass
    finally:
        argparse.ArgumentParser.parse_args = orig
        sys.argv, sys.path[:] = old_argv, old_path
        for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
            del sys.modules[m]
    return captured["ref"], captured["ours"]


def _table(parser):
    out = {}
    for a in parser._actions:
        for s in a.option_strings:
            out[s] = (a.dest, a.default, getattr(a.type, "__name__", a.type), a.choices, a.nargs, type(a).__name__, a.required)
    return out


def test_every_reference_option_is_taken_with_the_same_meaning():
    ref, ours = (_table(p) for p in _parsers())
    assert len(ref) >= 30
    differing = {k for k in set(ref) | set(ours) if ref.get(k) != ours.get(k)}
    assert differing == {"-t", "--threads"}, {k: (ref.get(k), ours.get(k)) for k in differing}
    assert ours["--threads"][1] == 1 and ours["--threads"][0] == ref["--threads"][0] and ours["--threads"][2] == ref["--threads"][2]
