#!/usr/bin/env python3
"""sha1 over the sources that decide what the GPU does: every file of porechop_amd/csrc/ except the host I/O code (pc_io.cpp,
pc_gz.h: FASTQ / gzip in and out, no launch in them) -- the kernels, the headers they share, the run-time kernel source, the
launch planning of pc_api.cpp / pc_jit.cpp, the Makefile's flags.

A profiles/*_summary.json is bound to the library it was taken with (its sha1).  A change to the host I/O code alone gives a
new library whose kernels, launches and HBM traffic are the same: bench.py therefore also accepts a summary whose
`device_sources_sha1` equals this fingerprint of the working tree.  The summary gets that field from
    python tools/device_fingerprint.py --stamp profiles/<round>_summary.json <commit>
which computes the fingerprint from the blobs of <commit> -- the commit the profiled library was built from.
    python tools/device_fingerprint.py            prints the working tree's fingerprint."""
import hashlib
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("porechop_amd", "csrc")
HOST_IO = {"pc_io.cpp", "pc_gz.h"}


def _wanted(name):
    return (name.endswith((".hip", ".h", ".cpp")) or name == "Makefile") and name not in HOST_IO


def fingerprint(commit=None):
    h = hashlib.sha1()
    if commit is None:
        names = sorted(n for n in os.listdir(os.path.join(REPO, CSRC)) if _wanted(n))
        blobs = [(n, open(os.path.join(REPO, CSRC, n), "rb").read()) for n in names]
    else:
        listed = subprocess.run(["git", "-C", REPO, "ls-tree", "--name-only", commit, CSRC + "/"], capture_output=True, text=True, check=True).stdout.split()
        names = sorted(os.path.basename(p) for p in listed if _wanted(os.path.basename(p)))
        blobs = [(n, subprocess.run(["git", "-C", REPO, "show", "%s:%s/%s" % (commit, CSRC, n)], capture_output=True, check=True).stdout) for n in names]
    for n, b in blobs:
        h.update(n.encode() + b"\0" + str(len(b)).encode() + b"\0" + b)
    return h.hexdigest()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--stamp":
        path, commit = sys.argv[2], sys.argv[3]
        with open(path) as f:
            sj = json.load(f)
        sj["device_sources_sha1"] = fingerprint(commit)
        sj["device_sources_commit"] = subprocess.run(["git", "-C", REPO, "rev-parse", commit], capture_output=True, text=True, check=True).stdout.strip()
        with open(path, "w") as f:
            json.dump(sj, f, indent=1)
        print(path, "device_sources_sha1", sj["device_sources_sha1"], "from", sj["device_sources_commit"][:10])
    else:
        print(fingerprint(sys.argv[1] if len(sys.argv) > 1 else None))
