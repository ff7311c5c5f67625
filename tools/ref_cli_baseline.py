#!/usr/bin/env python3
"""BASELINE.md's B1 -- the UNMODIFIED reference CLI with --threads -- timed in the BUILD container (the
GPU box has no /root/reference), beside B2 as bench.py measures it (worker processes over the compiled
reference .so driving the reference's per-read logic), on the same synthetic reads and the same cores.
What it shows: how far below its own C++ core the shipped CLI sits (its --threads pool is a GIL-bound
multiprocessing.dummy pool), i.e. that bench.py's cpu_baseline is an UPPER bound of the CLI.
    python tools/ref_cli_baseline.py [n_reads] [threads]        -> one JSON line"""
import json, os, shutil, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    from oracle.oracle import build_ref
    from porechop_amd.synth import make_reads

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    KEEP = os.environ.get("PC_B1_KEEP")
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    REFERENCE = "/root/reference"
    so = build_ref()
    tmp = tempfile.mkdtemp(prefix="pc_b1_")
    shutil.copytree(os.path.join(REFERENCE, "porechop"), os.path.join(tmp, "porechop"),
                    ignore=shutil.ignore_patterns("include", "src", "*.so", "__pycache__"))
    shutil.copy(so, os.path.join(tmp, "porechop", "cpp_functions.so"))
    shutil.copy(os.path.join(REFERENCE, "porechop-runner.py"), tmp)
    # the benchmark's reads (BASELINE configs[3]: 8 kb, adapters at 90 % / 50 % of the ends, 1 % chimeras)
    reads = make_reads(n, 8000, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=0.01, device="cpu")
    host = reads.arena[: n * 8000].numpy().tobytes().decode("ascii")
    fq = os.path.join(tmp, "reads.fastq")
    with open(fq, "w") as f:
        for i in range(n):
            f.write("@r%d\n%s\n+\n%s\n" % (i, host[i * 8000:(i + 1) * 8000], "I" * 8000))
    out = {"reads": n, "read_len": 8000, "cores": os.cpu_count(), "where": "build container (no GPU)"}
    for t in sorted({1, threads}):
        t0 = time.perf_counter()
        res = subprocess.run([sys.executable, os.path.join(tmp, "porechop-runner.py"), "-i", fq, "-o", os.path.join(tmp, "out_%d.fastq" % t),
                              "--threads", str(t), "-v", "0"], capture_output=True, text=True, cwd=tmp)
        dt = time.perf_counter() - t0
        assert res.returncode == 0, res.stderr[-2000:]
        out["B1_cli_threads_%d" % t] = {"seconds": dt, "reads_per_s": n / dt}
    # B2 as in bench.py: the same phases B + C per read through the compiled reference, one worker process per core
    import bench
    from dataclasses import asdict
    from porechop_amd.pipeline import ScanParams
    from tests.cpu_worker import run_chunk
    sets = [(s.name, s.start, s.end) for s in bench.load_panel_sets()]
    names = [s[0] for s in sets]
    matching = [names.index("SQK-NSK007"), names.index("1D^2 part 2")]     # what phase A finds on these reads
    seqs = [host[i * 8000:(i + 1) * 8000] for i in range(n)]
    done, dt, _ = bench.cpu_sample(run_chunk, lambda c: (c, sets, matching, asdict(ScanParams()), True), seqs, 1e9, threads)
    out["B2_process_pool_threads_%d" % threads] = {"seconds": dt, "reads_per_s": done / dt, "reads": done}
    out["B2_over_B1"] = out["B2_process_pool_threads_%d" % threads]["reads_per_s"] / out["B1_cli_threads_%d" % threads]["reads_per_s"]
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":       # (the B2 pool spawns workers, which re-import this file)
    main()
