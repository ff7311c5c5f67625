#!/usr/bin/env python3
"""Replay the differential-fuzz corpus through the runner ON THE GPU (no reference needed there): every case is regenerated
from its seed (tests/fuzzcase.py), its content checked against the recorded md5, run through porechop_amd.runner over the
HIP library by the route the case names (middle scan behind the exact prefilter or the score bound; whole file or a stream
of small blocks), and every output file must have the content the unchanged reference CLI produced in the build container
(`tools/diff_fuzz.py N seed --emit DIR` wrote the md5s).
    python tools/replay_fuzz.py CASES.json [first [count]]"""
import json
import os
import shutil
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import porechop_amd  # noqa: E402
from porechop_amd import io as pio, runner  # noqa: E402
from tests import readgen  # noqa: E402
from tests.fuzzcase import content_md5, make_case  # noqa: E402
from tests.runner_cases import options_from_argv  # noqa: E402


def replay(cases, verbose=True):
    """-> (replayed, mismatches, by_route)"""
    bad = 0
    routes = {}
    for k, c in enumerate(cases):
        work = tempfile.mkdtemp(prefix="pc_replay_")
        try:
            case = make_case(c["cseed"], work, sized_gzip=pio.gzip_file)
            assert case["mode"] == c["mode"] and case["argv"] == c["argv"], "case %d: the recipe drifted (%r)" % (k, c["cseed"])
            assert content_md5(case["input"]) == c["content_md5"], "case %d: regenerated input differs (%r)" % (k, c["cseed"])
            opts = options_from_argv(c["argv"])
            target = os.path.join(work, "got", "bins" if c["mode"] == "b" else c["mode"][2:])
            os.makedirs(os.path.dirname(target))
            porechop_amd.Aligner.fast_prefilter = bool(c["prefilter"])
            os.environ.pop("PC_STREAM_BLOCK_BYTES", None)
            if c["blocks"]:
                os.environ["PC_STREAM_BLOCK_BYTES"] = c["blocks"]
            try:
                runner.run(case["input"], barcode_dir=target if c["mode"] == "b" else None,
                           output=None if c["mode"] == "b" else target, options=opts)
                got, gexit = (readgen.output_md5s(target) if os.path.exists(target) else {}), None
            except runner.UsageError as e:
                got, gexit = {}, str(e)
            except ValueError as e:                           # (what the loader raises for files it cannot read)
                got, gexit = {}, "error: " + str(e)
            if c["exit"] == "traceback" and gexit is not None:
                gexit = "traceback"                            # the reference has no message to compare: failing is what counts
            ok = got == c["outputs"] and gexit == c["exit"]
            key = ("prefilter" if c["prefilter"] else "bound") + ("+streamed" if c["blocks"] else "") + (" gz" if c["input"].endswith(".gz") else "") + \
                (" dir" if c["input"] == "indir" else "")
            routes[key] = routes.get(key, 0) + 1
            bad += not ok
            if not ok and verbose:
                print("BAD case %d seed %d %s %s\n   want %r %r\n   got  %r %r" % (k, c["cseed"], c["mode"], " ".join(c["argv"]), c["exit"], c["outputs"], gexit, got), flush=True)
        finally:
            porechop_amd.Aligner.fast_prefilter = True
            os.environ.pop("PC_STREAM_BLOCK_BYTES", None)
            shutil.rmtree(work, ignore_errors=True)
    return len(cases), bad, routes


if __name__ == "__main__":
    import gzip
    with (gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])) as f:
        cases = json.load(f)
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else len(cases)
    import torch
    n, bad, routes = replay(cases[first:first + count])
    print("device=%s library=%s" % (torch.cuda.get_device_name(0), porechop_amd.load_library().pc_version().decode()))
    print("routes: " + ", ".join("%s %d" % kv for kv in sorted(routes.items())))
    print("cases=%d mismatches=%d" % (n, bad))
    sys.exit(1 if bad else 0)
