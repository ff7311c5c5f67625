#!/usr/bin/env python3
"""INTEGRATION.md mode A measured (GPU box): the reference's own, unchanged Python with ONLY porechop/cpp_functions.so
swapped for libporechop_amd.so -- every adapter_alignment() call is a single-pair GPU launch (nothing prefetches) -- on the
first N of the benchmark's reads, beside the same CLI over its own library; output md5s must agree.
    python tools/mode_a_rate.py [reads=300] [threads=1]"""
import hashlib, json, os, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import bench  # noqa: E402
import porechop_amd  # noqa: E402
from porechop_amd.synth import make_reads  # noqa: E402
from tests.ref_cli import DEFAULT_STAGE  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
threads = sys.argv[2] if len(sys.argv) > 2 else "1"
work = tempfile.mkdtemp(prefix="pc_mode_a_")
try:
    reads = make_reads(max(n, 1000), 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, device="cuda")
    fq = os.path.join(work, "reads.fastq")
    bench.write_fastq(reads, n, fq)
    del reads
    stage_a = os.path.join(work, "stage_a")
    shutil.copytree(DEFAULT_STAGE, stage_a, ignore=shutil.ignore_patterns("test", "__pycache__", "cpp_functions.so"))
    shutil.copy(porechop_amd.LIB_PATH, os.path.join(stage_a, "porechop", "cpp_functions.so"))
    out = {"reads": n, "threads": threads}
    for tag, stage in (("reference", DEFAULT_STAGE), ("mode_a", stage_a)):
        o = os.path.join(work, tag + ".fastq")
        t0 = time.perf_counter()
        res = subprocess.run([sys.executable, os.path.join(stage, "porechop-runner.py"), "-i", fq, "-o", o, "-v", "0", "--threads", threads],
                             capture_output=True, text=True, cwd=stage)
        dt = time.perf_counter() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        out[tag] = {"seconds": dt, "reads_per_s": n / dt, "md5": hashlib.md5(open(o, "rb").read()).hexdigest()}
    out["md5_equal"] = out["reference"]["md5"] == out["mode_a"]["md5"]
    print(json.dumps(out))
finally:
    shutil.rmtree(work, ignore_errors=True)
