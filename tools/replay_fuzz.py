#!/usr/bin/env python3
"""Replay cases emitted by `tools/diff_fuzz.py --emit DIR` through the runner ON THE GPU (no reference
needed): every output file must have the content the reference CLI produced in the build container.
    python tools/replay_fuzz.py DIR"""
import json
import os
import sys
import tempfile

sys.path.insert(0, ".")
from porechop_amd import runner
from tests import readgen
from tests.runner_cases import options_from_argv

d = sys.argv[1]
cases = json.load(open(os.path.join(d, "cases.json")))
bad = 0
for k, c in enumerate(cases):
    opts = options_from_argv(c["argv"])
    work = tempfile.mkdtemp(prefix="pc_replay_")
    target = os.path.join(work, "bins" if c["mode"] == "b" else c["mode"][2:])
    try:
        runner.run(os.path.join(d, c["input"]), barcode_dir=target if c["mode"] == "b" else None,
                   output=None if c["mode"] == "b" else target, options=opts)
        got, gexit = (readgen.output_md5s(target) if os.path.exists(target) else {}), None
    except runner.UsageError as e:
        got, gexit = {}, str(e)
    ok = got == c["outputs"] and gexit == c["exit"]
    bad += not ok
    if not ok:
        print("BAD case %d %s %s\n   want %r %r\n   got  %r %r" % (k, c["mode"], " ".join(c["argv"]), c["exit"], c["outputs"], gexit, got))
print("replayed=%d mismatches=%d" % (len(cases), bad))
