"""porechop_amd.dropin under the UNCHANGED reference Python (only where /root/reference exists):
the reference CLI runs over its own fixtures with the three phase drivers wrapped; the backend is
the CPU oracle here (no GPU in this container -- the GPU backend's strings are proven identical
to the oracle's by tests/test_gpu_parity.py).  Checks: byte-identical output files (md5 from the
goldens = the compiled reference's own output) and that the prefetch logic anticipated EVERY call
(zero memo misses), i.e. the batching changes no decision and leaves nothing to per-call launches."""
import hashlib
import io
import os
import shutil
import sys
import tempfile
from contextlib import redirect_stderr, redirect_stdout

import pytest

REFERENCE = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference checkout not present")


class OracleBackend:
    def __init__(self, oracle):
        self.oracle = oracle
        self.calls = 0

    def align(self, pairs, scores):
        self.calls += 1
        return [self.oracle.adapter_alignment(r, a, tuple(scores)) for r, a in pairs]


class OracleProductBackend(OracleBackend):
    """... with the product entry point the GPU backend has: every read x every adapter, read-major."""

    def align_product(self, reads, adapters, scores):
        self.calls += 1
        return [self.oracle.adapter_alignment(r, a, tuple(scores)) for r in reads for a in adapters]


def md5_of(path):
    h = hashlib.md5()
    if os.path.isdir(path):
        for fn in sorted(os.listdir(path)):
            h.update(fn.encode())
            with open(os.path.join(path, fn), "rb") as f:
                h.update(f.read())
    else:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


@pytest.fixture(scope="module")
def staged_reference():
    from oracle.oracle import build_ref
    so = build_ref()
    tmp = tempfile.mkdtemp(prefix="pc_dropin_")
    shutil.copytree(os.path.join(REFERENCE, "porechop"), os.path.join(tmp, "porechop"),
                    ignore=shutil.ignore_patterns("include", "src", "*.so", "__pycache__"))
    shutil.copy(so, os.path.join(tmp, "porechop", "cpp_functions.so"))   # only so that the wrapper imports
    sys.path.insert(0, tmp)
    for m in [k for k in sys.modules if k == "porechop" or k.startswith("porechop.")]:
        del sys.modules[m]
    import porechop.porechop as pp
    yield pp, tmp
    sys.path.remove(tmp)
    shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("backend_cls", [OracleBackend, OracleProductBackend])
@pytest.mark.parametrize("run", ["one_default", "one_mid97", "one_nosplit", "two_default", "barcodes_default",
                                 "albacore_mid85"])
def test_unchanged_reference_with_prefetching_dropin(staged_reference, goldens, oracle, run, backend_cls):
    pp, tmp = staged_reference
    import porechop.adapters as adapters_mod
    import porechop_amd.dropin as dropin
    info = goldens["runs"][run]
    for a in adapters_mod.ADAPTERS:            # fresh-process state
        a.best_start_score, a.best_end_score = 0.0, 0.0
    backend = backend_cls(oracle)
    st = dropin.install(pp, backend)
    out = os.path.join(tmp, "out_" + run)
    argv = ["porechop", "-i", os.path.join(REFERENCE, "test", info["fixture"]), "-v", "0"]
    tail = [out if t == "BARCODE_DIR" else t for t in info["argv_tail"]]
    if "-b" in tail:
        target = out
    else:
        target = out + ".fastq"
        argv += ["-o", target]
    argv += tail
    if "--threads" not in tail:
        argv += ["--threads", "1"]
    old = sys.argv
    sys.argv = argv
    try:
        buf = io.StringIO()
        with redirect_stdout(buf), redirect_stderr(buf):
            pp.main()
    finally:
        sys.argv = old
    assert md5_of(target) == info["output_md5"]
    s = dropin.stats()
    assert s["misses"] == 0, s
    assert s["hits"] == info["calls"], (s, info["calls"])
    assert backend.calls <= 64        # a handful of large batches instead of thousands of calls
