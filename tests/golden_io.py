"""Readers for the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import gzip
import hashlib
import json
import os
import random

from tests.pairgen import LINEAR_SCHEMES, SCHEMES, case_stream

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_ref_calls():
    """-> dict(meta, runs, strings, calls=[[read_idx, adapter_idx, [m,x,go,ge], result], ...])"""
    with gzip.open(os.path.join(GOLDEN_DIR, "ref_calls.json.gz"), "rt") as f:
        return json.load(f)


def load_panel():
    with open(os.path.join(GOLDEN_DIR, "panel.json")) as f:
        return json.load(f)


def load_synthetic():
    """Yields (read, adapter, scheme, reference_result); inputs are regenerated from the
    stored seed and verified against the stored sha1."""
    with gzip.open(os.path.join(GOLDEN_DIR, "ref_synthetic.json.gz"), "rt") as f:
        d = json.load(f)
    out = []
    for s in d["sets"]:
        rng = random.Random(s["scheme_seed"])
        schemes = LINEAR_SCHEMES if s.get("schemes") == "linear" else SCHEMES
        h = hashlib.sha1()
        for (rd, ad), res in zip(case_stream(s["seed"], s["count"]), s["results"]):
            sc = rng.choice(schemes)
            h.update(("%s|%s|%r\n" % (rd, ad, sc)).encode())
            out.append((rd, ad, sc, res))
        assert h.hexdigest() == s["inputs_sha1"], "tests/pairgen.py drifted from the goldens"
    return out


def comparable(result):
    """The reference leaves every field but #0 (and the score) uninitialised when it reports
    failure (read_start == -1, alignment.cpp:9-21): compare only what it defines."""
    f = result.split(",")
    if f[0] == "-1":
        return ("-1", f[4])
    return tuple(f)
