"""The host ingest / output / gzip code (porechop_amd/csrc/pc_io.cpp + pc_gz.h) under AddressSanitizer
and UndefinedBehaviorSanitizer: tests/host/fuzz_io.cpp writes an irregular FASTQ file, compresses it in the three layouts the
readers distinguish (sized members, ONE member, `cat`-ed members with zero padding between two of them), reads each back whole
and as a stream of blocks, writes the reads out plain and compressed and reads those back -- all compared with the plain file's
reads -- and then pushes damaged copies (cut short, flipped bytes, zeroed ranges, garbage appended, chunks duplicated or
removed) through every reader: any return code is accepted, a sanitizer report or a crash is not.  (A few rounds here; 240 rounds
over four more seeds are recorded in profiles/r05_fuzz_io.txt.)

Found this way: zlib's gzread, which the whole-file route used, silently stops at zero padding between members (Python's
gzip module -- the reference's reader, porechop/misc.py:60-81 -- skips it) and returns a truncated stream's bytes without an
error (the reference raises); the whole-file route now drains the streamed route's producer instead."""
# (ThreadSanitizer does not start in this image; the one-shot route's watching of another thread's output is by design a
# race in C++ terms -- see oneshot_member in pc_io.cpp for why it is sound on x86.)
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ingest_output_and_gzip_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = tmp_path / "fuzz_io"
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", os.path.join(REPO, "porechop_amd", "csrc", "pc_io.cpp"),
                            os.path.join(REPO, "tests", "host", "fuzz_io.cpp"), "-o", str(exe), "-lz", "-ldl", "-lpthread"],
                           capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and ("asan" in build.stderr or "ubsan" in build.stderr):
        pytest.skip("this g++ has no sanitizer runtimes: " + build.stderr[-300:])
    assert build.returncode == 0, build.stderr[-3000:]
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    # the third and fourth: every member the workers do not take goes through the watched one-shot libdeflate route
    # (oneshot_member), with room for it and with a room it outgrows after ~5 MB were handed over (seed 1's file is 6.1 MB)
    configs = ((1, 1, {}), (2, 1, {"PC_NO_LIBDEFLATE": "1", "PC_GZ_SPEC_CAP_MB": "1"}),
               (1, 1, {"PC_GZ_ONESHOT_MIN_MB": "0", "PC_GZ_VERBOSE": "1"}),
               (1, 1, {"PC_GZ_ONESHOT_MIN_MB": "0", "PC_GZ_ONESHOT_ROOM_KB": "5700", "PC_GZ_VERBOSE": "1"}))
    # (the four runs side by side, each in a directory of its own: they share nothing but the binary)
    procs = []
    for k, (seed, rounds, extra) in enumerate(configs):
        wk = work / ("run%d" % k)
        wk.mkdir()
        procs.append(subprocess.Popen([str(exe), str(wk), str(seed), str(rounds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=dict(env, **extra)))
    for (seed, rounds, extra), pr in zip(configs, procs):
        try:
            out, err = pr.communicate(timeout=1200)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert pr.returncode == 0, (seed, extra, out[-1500:], err[-6000:])
        assert "0 check failure(s)" in out and "ERROR: AddressSanitizer" not in err and "runtime error" not in err, \
            (out[-1500:], err[-6000:])
        if "PC_GZ_ONESHOT_ROOM_KB" in extra:
            assert "result 2" in err, err[-2000:]          # out of room after a hand-over: zlib restarted, discarding it
        elif "PC_GZ_ONESHOT_MIN_MB" in extra:
            assert "result 1" in err, err[-2000:]
