#!/usr/bin/env python3
"""bench.py -- end + middle adapter scan of synthetic 8 kb reads on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over this rank's batch of reads, inputs already resident in
HBM: phase A (adapter-set presence over the check reads, 119-set panel, MAX all-reduce across
ranks -- the only collective), phase B (end windows vs the matching sets -> trim amounts) and
phase C (whole trimmed reads vs the matching sets' adapters, including the sequential
mask-and-realign rounds for reads with middle hits).  Headline workload = BASELINE.json configs[3]
("1M synthetic 8 kb reads with 1% chimeras, middle scan enabled") per GPU; reads are sharded
over ranks with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0, under 8 KB, numbers only (DESIGN.md section 9 names every key; the unabridged record goes
to --full-json, default gpurun_out/bench_full.json):
  roofline     -- the dominant kernel (score-only whole-read scan), timed with HIP events on its launch stream inside the
                  timed region (pc_get_timing); traffic from profiles/<round>_summary.json when that profile was taken with
                  the library now loaded (traffic_measured_in_this_run is always false)
  cpu_baseline -- phases B+C of a bounded sample of the same reads on the host cores through the compiled reference
                  (oracle/_ref; BASELINE.md's B2), and b1_cli*: Porechop's OWN --threads CLI, unchanged (tests/ref_cli.py runs
                  the staged porechop.porechop.main()), on the first --cli-reads of the same reads, --threads 1 and 16
  parity       -- the CPU sample's per-read results vs the GPU's; phase A re-derived on the CPU; int16 vs fp16 kernels over ALL
                  pairs; the batch runner's output file vs the reference CLI's on the same reads
  dropin       -- the unchanged reference Python + porechop_amd.dropin over the HIP library: md5 vs the CPU reference, memo
                  misses, reads/s on --dropin-reads reads
  legs (+ flat config.<leg>_* copies; N=1 only, except configs4_per_gpu which every rank runs):
                  configs1 (100 k reads, end-trim only), configs2 (1 M barcoded reads: every end-window pair traced, and
                  pruned_* = phase B pruned exactly), configs4_per_gpu (1.25 M reads, barcodes + chimeras + 196 middle adapters:
                  every record, and fast_* = exact prefilter + pruned phase B), exact_prefilter (the headline step behind the
                  prefilter), proven_middle_scan, ragged_lengths, from_host_memory (2 bits per base over PCIe, steady state;
                  unpacked_* = 1 B/base), end_to_end (FASTQ file -> FASTQ file)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# VALU issue ceiling (tools/ubench_valu.hip, profiles/r02_ubench_valu.txt): one wave64 VALU instruction
# per 4 shader cycles per SIMD (16 lanes/clk/SIMD); 1024 SIMDs at the 2.4 GHz peak clock
VALU_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 4.0
HBM_PEAK_GBS = 8000.0


def load_panel_json():
    with open(os.path.join(REPO, "tests", "golden", "panel.json")) as f:
        return json.load(f)


def load_panel_sets():
    from porechop_amd.pipeline import AdapterSet
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None,
                       tuple(a["end"]) if a["end"] else None) for a in load_panel_json()]


def host_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota, not os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def one_step(pl, reads, n_check, world, proofs=False, prefilter=False):
    """The hot path over one resident batch.  Returns (matching, start_trim, end_trim, hits).
    proofs=True is the optional variant with the two exact prunings of DESIGN.md section 7 (f-4 and
    the proven middle scan), prefilter=True the one with the exact bit-parallel prefilter in front of the middle
    scan; the headline measurement uses neither."""
    from porechop_amd.distributed import reduce_presence
    check = None if n_check >= reads.n else torch.arange(n_check, device=reads.off.device)
    best_s, best_e = pl.phase_a(reads, check, prune=proofs)
    # adapter-set presence is the one cross-read reduction (porechop.py:327): 119 x 2 maxima, MAX
    # all-reduce over RCCL (a no-op at world size 1)
    best_s, best_e = reduce_presence(best_s, best_e)
    matching = pl.matching_sets(best_s, best_e)
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching, prove=proofs, prefilter=prefilter)
    return matching, st, et, hits


def step_end_trim(pl, reads, n_check):
    """BASELINE configs[1] (--no_split): phases A + B."""
    bs, be = pl.phase_a(reads, torch.arange(min(n_check, reads.n), device=reads.off.device))
    matching = pl.matching_sets(bs, be)
    st, et = pl.phase_b(reads, matching)
    return matching, st, et


def step_demux(pl, reads, n_check, opts, prune=False):
    # prune=False: every end-window pair traced (the number earlier rounds reported); True: exact pruning (pc_select.hip)
    """BASELINE configs[2] (-b DIR): phase A, the barcode-kit choice (porechop.py:330-371), phase B
    with the barcode identities, determine_barcode (nanopore_read.py:399-466) for every read."""
    from porechop_amd import panel as rules
    from porechop_amd.runner import barcode_bins
    bs, be = pl.phase_a(reads, torch.arange(min(n_check, reads.n), device=reads.off.device))
    matching = pl.matching_sets(bs, be)
    bsh, beh = bs.cpu().numpy(), be.cpu().numpy()
    index_of = {id(s): i for i, s in enumerate(pl.sets)}
    orientation = rules.choose_barcoding_kit([pl.sets[i] for i in matching], lambda s: bsh[index_of[id(s)]],
                                             lambda s: beh[index_of[id(s)]])
    bc_sets = [i for i in matching if rules.is_barcode(pl.sets[i]) and rules.barcode_direction(pl.sets[i]) == orientation]
    names, bins = barcode_bins(pl, bc_sets)
    st, et, calls = pl.phase_b_demux(reads, matching, bins, opts.barcode_threshold, opts.barcode_diff, opts.require_two_barcodes, prune=prune)
    return matching, orientation, names, st, et, calls


def host_seqs(reads, n):
    ln = int(reads.length[0].item())
    host = reads.arena[: n * ln].cpu().numpy().tobytes().decode("ascii")
    return [host[i * ln:(i + 1) * ln] for i in range(n)], ln


def cpu_sample(worker, make_args, seqs, seconds, workers, probe=4):
    """Run `worker` (tests/cpu_worker.py) over a bounded sample of `seqs` on `workers` spawned
    processes, sized from a probe so that the leg takes about `seconds`.
    -> (reads done, wall seconds, per-read results in read order)."""
    import multiprocessing as mp
    _, t_probe, _ = worker(make_args(seqs[:probe]))
    per_read = t_probe / probe
    n = int(min(len(seqs), max(workers * 2, seconds * workers / max(per_read, 1e-6))))
    per = max(1, n // workers)
    chunks = [seqs[i:i + per] for i in range(0, n, per)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        pool.map(worker, [make_args(c[:1]) for c in chunks])     # start-up + import, untimed
        t0 = time.perf_counter()
        res = pool.map(worker, [make_args(c) for c in chunks])
        dt = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    per_read_results = [x for r in res for x in r[2]]
    return done, dt, per_read_results


def baseline_kind():
    from oracle.oracle import REF_SO
    return "reference" if os.path.isfile(REF_SO) else "port"


def kind_text(kind):
    return ("compiled reference oracle/_ref/cpp_functions.so" if kind == "reference" else "oracle port oracle/pc_oracle.c")


def cpu_baseline(reads, pl, matching, st, et, hits, seconds, workers):
    """Phases B + C of the same reads, the reference's sequential per-read logic
    (tests/ref_pipeline.py), on all host cores: one spawned worker PROCESS per core over the compiled
    reference -- BASELINE.md's "B2" (Porechop's own --threads pool is GIL-bound, README.md:355-359;
    processes are its fair upper bound).  The sample's results are compared with the GPU's."""
    from dataclasses import asdict
    from tests.cpu_worker import run_chunk
    kind = baseline_kind()
    seqs, ln = host_seqs(reads, min(reads.n, 49152))       # (the probe sizes the sample to about --cpu-seconds of wall clock)
    sets = [(s.name, s.start, s.end) for s in pl.sets]
    params = asdict(pl.p)
    done, dt, res = cpu_sample(run_chunk, lambda c: (c, sets, matching, params, True), seqs, seconds, workers)
    got = {}
    for r, a, s, e in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
        if r < done:
            got.setdefault(r, []).append((a, s, e))
    stl, etl = st[:done].cpu().tolist(), et[:done].cpu().tolist()
    bad = [r for r in range(done) if (stl[r], etl[r], got.get(r, [])) != (res[r][0], res[r][1], list(res[r][2]))]
    base = {"value": done / dt, "unit": "reads/s", "cores": workers, "kind": kind,
            "label": "%s, process pool (BASELINE.md B2: fair upper bound of the reference's --threads CLI)" % kind_text(kind),
            "sample": "%d of the benchmark's reads (%d bp each), phases B+C, %d worker processes over the %s, %.1f s wall"
                      % (done, ln, workers, kind_text(kind), dt)}
    parity = {"checked": done, "mismatches": len(bad), "what": "start trim, end trim, middle hits (adapter, start, end) per read",
              "first_mismatching_reads": bad[:8]}
    return base, parity


def write_fastq(reads, n, path, first=0):
    """Reads first .. first + n of a uniform-length resident batch as a plain 4-line FASTQ file (names r<index>,
    qualities '5': SURVEY.md 8d)."""
    L = int(reads.length[0].item())
    seq = reads.arena[first * L:(first + n) * L].view(n, L).cpu().numpy()
    name_w = 9
    rec = np.empty((n, 1 + name_w + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
    rec[:, 2:1 + name_w] = (np.arange(first, first + n)[:, None] // (10 ** np.arange(name_w - 2, -1, -1))[None, :]) % 10 + ord("0")
    c = 1 + name_w
    rec[:, c] = 10
    rec[:, c + 1:c + 1 + L] = seq
    rec[:, c + 1 + L:c + 4 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, c + 4 + L:c + 4 + 2 * L] = ord("5")
    rec[:, -1] = 10
    rec.tofile(path)
    return int(rec.size)


def file_md5(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run_ref_cli(fastq, out, threads, dropin=False, extra=()):
    """tests/ref_cli.py in a fresh interpreter: the staged, unchanged porechop.porechop.main() (what porechop-runner.py
    calls), phases timed from outside; dropin=True installs porechop_amd.dropin first (INTEGRATION.md mode B).
    -> (report dict, wall seconds of the whole process incl. interpreter start)."""
    import subprocess
    import tempfile
    rep = tempfile.mktemp(prefix="pc_cli_", suffix=".json")
    cmd = [sys.executable, os.path.join(REPO, "tests", "ref_cli.py")] + (["--dropin"] if dropin else []) + \
          ["--report", rep, "--", "-i", fastq, "-o", out, "--threads", str(threads), "-v", "0"] + list(extra)
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if res.returncode != 0:
        raise RuntimeError("ref_cli.py failed: " + res.stderr[-1500:])
    with open(rep) as f:
        report = json.load(f)
    os.remove(rep)
    return report, wall


def leg_reference_cli(reads, args, workers, work):
    """BASELINE.md's B1 -- Porechop's OWN --threads CLI (porechop/porechop.py:86,108,484-509,575-591), unchanged, over its
    own compiled cpp_functions.so, on the first --cli-reads of the benchmark's reads written as FASTQ, on this box's host
    cores, in this run: --threads 1 and --threads min(cores, 16) (the CLI's default cap)."""
    n = min(args.cli_reads, reads.n)
    fq = os.path.join(work, "cli_reads.fastq")
    write_fastq(reads, n, fq)
    out = {"kind": "reference CLI", "reads": n, "read_len": int(reads.length[0].item())}
    best = 0.0
    for t in sorted({1, min(workers, 16)}):
        o = os.path.join(work, "cli_out_t%d.fastq" % t)
        rep, wall = run_ref_cli(fq, o, t)
        out["threads_%d" % t] = {"reads_per_s": n / rep["main_s"], "main_s": rep["main_s"], "process_s": wall,
                                 "phase_s": {k: round(v, 3) for k, v in rep["phase_s"].items()}}
        out["output_md5"] = file_md5(o) if "output_md5" not in out else out["output_md5"]
        out["same_output_all_threads"] = out.get("same_output_all_threads", True) and file_md5(o) == out["output_md5"]
        best = max(best, n / rep["main_s"])
    out["best_reads_per_s"] = best
    return out, fq


def leg_dropin(reads, args, work, cli, cli_fastq):
    """What a Porechop user gets from the drop-in: the unchanged reference Python + porechop_amd.dropin.install(pp)
    (INTEGRATION.md mode B) with the real library.  (1) the reference CLI leg's FASTQ again: output md5 must equal the
    CPU reference's, zero memo misses; (2) --dropin-reads reads for the rate (the reference's own Python loops bound it)."""
    out = {}
    o = os.path.join(work, "dropin_small.fastq")
    rep, _ = run_ref_cli(cli_fastq, o, 1, dropin=True)
    out["md5_equal"] = bool(cli and file_md5(o) == cli.get("output_md5"))
    out["md5_checked_reads"] = cli["reads"] if cli else 0
    out["misses"] = rep["dropin"]["misses"]
    n = min(args.dropin_reads, reads.n)
    fq = os.path.join(work, "dropin_reads.fastq")
    write_fastq(reads, n, fq)
    for t in sorted({1, args.dropin_threads}):
        o = os.path.join(work, "dropin_big_t%d.fastq" % t)
        rep, wall = run_ref_cli(fq, o, t, dropin=True)
        r = {"reads_per_s": n / rep["main_s"], "main_s": rep["main_s"], "process_s": wall,
             "phase_s": {k: round(v, 3) for k, v in rep["phase_s"].items()},
             "prefetch_s": round(rep["dropin"]["prefetch_s"], 3), "gpu_batch_calls_s": round(rep["dropin"]["backend_s"], 3),
             "lookups": rep["dropin"]["hits"], "misses": rep["dropin"]["misses"], "pairs_batched": rep["dropin"]["batched"]}
        out["threads_%d" % t] = r
        out["misses"] += rep["dropin"]["misses"]
        if r["reads_per_s"] > out.get("reads_per_s", 0.0):
            out["reads_per_s"], out["threads"] = r["reads_per_s"], t
    out["reads"] = n
    return out


def cpu_phase_a_check(reads, pl, workers, nreads=1024):
    """Phase A re-derived on the host for the first `nreads` reads (the reference's align_adapter_set over the whole
    119-set panel, nanopore_read.py:149-164, through the compiled reference) against the GPU's phase A over the same
    reads: every entry of both presence tables must be equal, and so must the matching sets -- the CPU parity legs
    then no longer depend on a `matching` list the GPU made."""
    import multiprocessing as mp
    from dataclasses import asdict
    from tests.cpu_worker import run_chunk_phase_a
    k = min(reads.n, nreads)
    seqs, _ = host_seqs(reads, k)
    sets = [(s.name, s.start, s.end) for s in pl.sets]
    per = max(1, (k + workers - 1) // workers)
    chunks = [seqs[i:i + per] for i in range(0, k, per)]
    with mp.get_context("spawn").Pool(min(workers, len(chunks))) as pool:
        res = pool.map(run_chunk_phase_a, [(c, sets, asdict(pl.p), True) for c in chunks])
    bs = np.max(np.array([r[2][0] for r in res]), axis=0)
    be = np.max(np.array([r[2][1] for r in res]), axis=0)
    gs, ge = pl.phase_a(reads, torch.arange(k, device=reads.off.device))
    m_gpu = pl.matching_sets(gs, ge)
    best = np.maximum(bs, be)
    m_cpu = [i for i, s in enumerate(pl.sets) if "(full sequence)" not in s.name and best[i] >= pl.p.adapter_threshold]
    gs, ge = gs.cpu().numpy(), ge.cpu().numpy()
    bad = int((gs != bs).sum() + (ge != be).sum())
    return {"reads": k, "table_entries": int(2 * len(sets)), "entries_differing": bad, "same_matching_sets": m_gpu == m_cpu,
            "matching_sets_cpu": [pl.sets[i].name for i in m_cpu]}


def device_crosscheck(pl, reads, matching, st, et, dev):
    """Full-coverage exactness evidence at benchmark size, outside every timed region: phase B's end-window records
    and phase C's whole-read records of ALL reads of the batch, once with the kernels the benchmark ran (packed fp16
    where the host gates of csrc/pc_bounds.h prove it exact) and once with the packed-int16 kernels
    (pc_set_int16_only), compared record for record on the device."""
    import hashlib
    from porechop_amd.batch import MODE_TRACE, MODE_TWO_PASS
    from porechop_amd.pipeline import trimmed_interval

    def records():
        jobs, _ = pl._phase_b_jobs(reads, matching)
        _, out_b, _ = pl._scan_jobs(reads.arena, jobs, MODE_TRACE, pl.p.end_size, with_layout=True)
        s_pos, e_pos = trimmed_interval(reads.length, st, et)
        tlen = torch.clamp(e_pos - s_pos, min=0).to(torch.int32)
        toff = reads.off + s_pos
        ads = pl._middle_adapters_with_sets(matching)
        jobs_c = [(pl.seq_index[a[1]], toff, tlen, ("set", si)) for a, si in ads]
        _, out_c, _ = pl._scan_jobs(reads.arena, jobs_c, MODE_TWO_PASS, int(tlen.max().item()), with_layout=True)
        pl.aligner.sync()
        return out_b, out_c

    b16, c16 = records()
    ops16 = pl.aligner.trace_ops_per_2_cells()
    pl.aligner.set_int16_only(True)
    try:
        bi, ci = records()
        opsi = pl.aligner.trace_ops_per_2_cells()
    finally:
        pl.aligner.set_int16_only(False)
    diff_b = int((b16 != bi).any(dim=1).sum().item())
    diff_c = int((c16 != ci).any(dim=1).sum().item())
    sha = hashlib.sha1(b16.cpu().numpy().tobytes())
    sha.update(c16.cpu().numpy().tobytes())
    return {"device_crosscheck_pairs": int(b16.shape[0] + c16.shape[0]), "end_window_pairs": int(b16.shape[0]),
            "whole_read_pairs": int(c16.shape[0]), "records_differing": diff_b + diff_c,
            "traced_kernel_ops_per_2_cells": [ops16, opsi], "records_sha1": sha.hexdigest(),
            "what": "every 8-int record of phases B and C of the whole headline batch: packed-fp16 kernels vs packed-int16 kernels"}


def trace_roofline(timing, n_windows, pairs, cells, steps, ops_per_2_cells, leg=None):
    """Roofline object of the traced end-window kernel from the library's per-launch HIP-event timing.
    Algorithmic bytes: every end window (150 B) in once, 28 B out per (window, adapter)."""
    ms, launches, tpairs = timing["trace"]
    if launches <= 0:
        return None
    per_launch_s = ms / 1e3 / launches
    alg_total = (n_windows * 150.0 + 28.0 * pairs) * steps
    achieved = alg_total / launches / per_launch_s / 1e9
    gcups = cells * steps / (ms / 1e3) / 1e9
    peak_gcups = VALU_WAVE_INSTR_PER_S * 64 * 2 / ops_per_2_cells / 1e9
    return {"bound": "valu", "kernel": "traced end-window scan (trace bits + on-device traceback/digest)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": profile_traffic(leg, "trace16_kernel", launches / max(1, steps)) if leg else None,
            "launches": int(launches), "avg_launch_ms": per_launch_s * 1e3,
            "algorithmic_bytes_per_launch": alg_total / launches,
            "valu": {"achieved_gcups": gcups, "peak_gcups": peak_gcups, "frac": gcups / peak_gcups,
                     "ops_per_2_cells": ops_per_2_cells},
            "note": "HBM fraction on ALGORITHMIC bytes (2*150 + 28*2P per read); the kernel is VALU-bound (>= 20 cells per "
                    "algorithmic byte) and its 4-bit/cell trace slab is implementation traffic (profiles/, DESIGN.md section 4)"}


VERBOSE = os.environ.get("PC_BENCH_VERBOSE", "0") not in ("", "0")


def note(msg):
    if VERBOSE:
        print("[bench %.3f] %s" % (time.perf_counter(), msg), file=sys.stderr, flush=True)


def timed(fn, steps, warmup, sync):
    for _ in range(warmup):
        out = fn()
        sync()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        out = fn()
        if VERBOSE:
            sync()
            note("step %d done at +%.1f ms" % (k, (time.perf_counter() - t0) * 1e3))
    sync()
    return out, time.perf_counter() - t0


def leg_configs1(dev, args, workers):
    """BASELINE configs[1]: 100 k synthetic 8 kb reads, SQK-LSK109-style ligation adapters, end-trim only."""
    from dataclasses import asdict
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads
    from tests.cpu_worker import run_chunk
    p = ScanParams()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    n = args.reads1
    reads = make_reads(n, args.read_len, seed=1, start_frac=0.9, end_frac=0.5, chimera_frac=0.0, device=dev)

    def sync():
        pl.aligner.sync()
        torch.cuda.synchronize()
    timed(lambda: step_end_trim(pl, reads, p.check_reads), 0, max(1, args.warmup), sync)
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    (matching, st, et), dt = timed(lambda: step_end_trim(pl, reads, p.check_reads), args.steps, 0, sync)
    timing = pl.aligner.get_timing()
    pl.aligner.set_timing(False)
    out = {"workload": "BASELINE configs[1]: %d synthetic %d-bp reads, end-trim only (--no_split): phases A (119-set panel, "
                       "%d check reads) + B" % (n, args.read_len, p.check_reads),
           "reads_per_s": n * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
           "matching_sets": [pl.sets[i].name for i in matching],
           "kernel_ms_per_step": {k: v[0] / args.steps for k, v in timing.items()}}
    # phase B alone (the end-scan kernel of north_star): its own roofline
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    _, dtb = timed(lambda: pl.phase_b(reads, matching), args.steps, 0, sync)
    tb = pl.aligner.get_timing()
    pl.aligner.set_timing(False)
    ads = [(pl.sets[i].start, pl.sets[i].end) for i in matching]
    pairs = n * sum((s is not None) + (e is not None) for s, e in ads)
    cells = n * 150 * sum((len(s[1]) if s else 0) + (len(e[1]) if e else 0) for s, e in ads)
    out["phase_b"] = {"reads_per_s": n * args.steps / dtb, "ms_per_step": dtb / args.steps * 1e3, "pairs_per_read": pairs / n}
    # the end-scan kernel of north_star = phase B alone: rate and launch time from THIS region; no counter traffic here -- the
    # profiled leg (profiles/<round>_summary.json) runs whole steps, phases A + B, and its counters belong to that scope:
    out["roofline"] = trace_roofline(tb, 2 * n, pairs, cells, args.steps, pl.aligner.trace_ops_per_2_cells(), leg=None)
    nchk = min(p.check_reads, n)
    pa = [(s.start, s.end) for s in pl.sets if "(full sequence)" not in s.name]
    pairs_a = nchk * sum((s is not None) + (e is not None) for s, e in pa)
    cells_a = nchk * 150 * sum((len(s[1]) if s else 0) + (len(e[1]) if e else 0) for s, e in pa)
    out["step_roofline"] = trace_roofline(timing, 2 * n + 2 * nchk, pairs + pairs_a, cells + cells_a, args.steps,
                                          pl.aligner.trace_ops_per_2_cells(), leg="configs1")
    if out["step_roofline"]:
        out["step_roofline"]["scope"] = "whole step: phases A + B, every traced launch (counter traffic and algorithmic bytes on this one scope)"
    if args.cpu_seconds > 0:
        seqs, ln = host_seqs(reads, min(n, 16384))
        sets = [(s.name, s.start, s.end) for s in pl.sets]
        done, dtc, res = cpu_sample(run_chunk, lambda c: (c, sets, matching, asdict(p), True, False), seqs,
                                    min(args.cpu_seconds, 4.0), workers, probe=64)
        stl, etl = st[:done].cpu().tolist(), et[:done].cpu().tolist()
        bad = [r for r in range(done) if (stl[r], etl[r]) != (res[r][0], res[r][1])]
        out["cpu_baseline"] = {"value": done / dtc, "unit": "reads/s", "cores": workers, "kind": baseline_kind(),
                               "sample": "%d reads, phase B, %d worker processes, %.2f s wall" % (done, workers, dtc)}
        out["parity"] = {"checked": done, "mismatches": len(bad), "what": "start trim, end trim per read"}
        out["speedup_vs_cpu_baseline"] = out["phase_b"]["reads_per_s"] / out["cpu_baseline"]["value"]
    pl.close()
    return out


def leg_configs2(dev, args, workers):
    """BASELINE configs[2]: 1 M synthetic 8 kb barcoded reads, full panel incl. the 96 barcodes, end-trim + demux."""
    from dataclasses import asdict
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.runner import Options
    from porechop_amd.synth import make_reads
    from tests.cpu_worker import run_chunk_barcodes
    p = ScanParams()
    opts = Options()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    n = args.reads2
    fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
    reads = make_reads(n, args.read_len, seed=2, start_frac=0.9, end_frac=0.5, chimera_frac=0.0, device=dev,
                       barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])

    def sync():
        pl.aligner.sync()
        torch.cuda.synchronize()
    timed(lambda: step_demux(pl, reads, p.check_reads, opts), 0, max(1, min(args.warmup, 2)), sync)
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    steps = max(1, min(args.steps, 10))
    (matching, orientation, names, st, et, calls), dt = timed(lambda: step_demux(pl, reads, p.check_reads, opts), steps, 0, sync)
    timing = pl.aligner.get_timing()
    # the same step with phase B pruned exactly: a score-only pass over every pair, then only the pairs that can
    # matter are traced (two selection rounds on the device)
    pruned = None
    if pl.can_prune_phase_b:
        timed(lambda: step_demux(pl, reads, p.check_reads, opts, prune=True), 0, 1, sync)
        pl.aligner.get_timing()
        pl.stats["pairs_end"] = pl.stats["pairs_end_traced_after_pruning"] = 0
        (_, _, _, st_p, et_p, calls_p), dt_p = timed(lambda: step_demux(pl, reads, p.check_reads, opts, prune=True), steps, 0, sync)
        timing_p = pl.aligner.get_timing()
        pruned = {"reads_per_s": n * steps / dt_p, "ms_per_step": dt_p / steps * 1e3, "steps": steps,
                  "speedup": dt / dt_p,
                  "same_trims_and_calls_as_tracing_every_pair": bool(torch.equal(st_p, st) and torch.equal(et_p, et) and np.array_equal(calls_p, calls)),
                  "pairs_traced_fraction": pl.stats["pairs_end_traced_after_pruning"] / max(1, pl.stats["pairs_end"]),
                  "kernel_ms_per_step": {k: v[0] / steps for k, v in timing_p.items() if v[1]},
                  "what": "phase B as score-only pass over every (read end, sequence) pair + exact selection of the pairs that can "
                          "change a trim or a call (bounds from the end cell and score; pc_select.hip) + traced scan of those; "
                          "tests/test_gpu_phase_b_pruning.py checks the bounds for EVERY pair of its batches"}
    pl.aligner.set_timing(False)
    ads = [(pl.sets[i].start, pl.sets[i].end) for i in matching]
    pairs = n * sum((s is not None) + (e is not None) for s, e in ads)
    cells = n * 150 * sum((len(s[1]) if s else 0) + (len(e[1]) if e else 0) for s, e in ads)
    # phase A's launches are in the timing too: its pairs/cells (check reads x whole panel)
    nchk = min(p.check_reads, n)
    pa = [(s.start, s.end) for s in pl.sets if "(full sequence)" not in s.name]
    pairs_a = nchk * sum((s is not None) + (e is not None) for s, e in pa)
    cells_a = nchk * 150 * sum((len(s[1]) if s else 0) + (len(e[1]) if e else 0) for s, e in pa)
    truth = reads.truth_barcode.cpu().numpy()
    want = np.array([names.index("BC%02d" % (b + 1)) if "BC%02d" % (b + 1) in names else -2 for b in range(len(fw))])[truth]
    out = {"workload": "BASELINE configs[2]: %d synthetic %d-bp reads, each with barcode b ~ U{1..96} (BCb after the start "
                       "adapter, BCb_rev before the end adapter), full 119-set panel, end-trim + demultiplexing: phases A + "
                       "kit choice + B (%d pairs per read) + barcode calls" % (n, args.read_len, pairs // n),
           "reads_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
           "matching_sets": len(matching), "matching_barcode_sets": sum(1 for i in matching if pl.sets[i].name.startswith("Barcode ")),
           "barcode_orientation": orientation, "pairs_per_read": pairs / n,
           "reads_binned_to_their_planted_barcode": float((calls == want).mean()), "reads_unassigned": float((calls < 0).mean()),
           "kernel_ms_per_step": {k: v[0] / steps for k, v in timing.items()},
           "roofline": trace_roofline(timing, 2 * n + 2 * nchk, pairs + pairs_a, cells + cells_a, steps,
                                      pl.aligner.trace_ops_per_2_cells(), leg="configs2")}
    if pruned:
        kms = pruned["kernel_ms_per_step"]
        score_ms = kms.get("score_spec", 0.0) + kms.get("score", 0.0)
        if score_ms > 0:
            pruned["score_pass"] = {"ms_per_step": score_ms, "cells_per_step": cells,
                                    "tcups": cells / (score_ms / 1e3) / 1e12,
                                    "packed_ops_per_2_cells": 5.0,
                                    "what": "score-only kernels (barcodes 2k-1 and 2k share a pass; launches alternate between two "
                                            "streams and are timed as one region per row class)"}
        out["exact_pruning"] = pruned
    if args.cpu_seconds > 0:
        seqs, ln = host_seqs(reads, min(n, 4096))
        sets = [(s.name, s.start, s.end) for s in pl.sets]
        mk = lambda c: (c, sets, matching, asdict(p), True, orientation, opts.barcode_threshold, opts.barcode_diff,
                        opts.require_two_barcodes)
        done, dtc, res = cpu_sample(run_chunk_barcodes, mk, seqs, min(args.cpu_seconds, 6.0), workers, probe=4)
        stl, etl = st[:done].cpu().tolist(), et[:done].cpu().tolist()
        cl = [names[k] if k >= 0 else "none" for k in calls[:done]]
        bad = [r for r in range(done) if (stl[r], etl[r], cl[r]) != tuple(res[r])]
        out["cpu_baseline"] = {"value": done / dtc, "unit": "reads/s", "cores": workers, "kind": baseline_kind(),
                               "sample": "%d reads, phase B + barcode call, %d worker processes, %.2f s wall" % (done, workers, dtc)}
        out["parity"] = {"checked": done, "mismatches": len(bad), "what": "start trim, end trim, barcode call per read",
                         "first_mismatching_reads": bad[:8]}
        out["speedup_vs_cpu_baseline"] = out["reads_per_s"] / out["cpu_baseline"]["value"]
    try:
        out.setdefault("parity", {})["device_crosscheck"] = crosscheck_end_windows(pl, reads, matching)
    except Exception as e:
        out.setdefault("parity", {})["device_crosscheck"] = {"failed": repr(e)}
    pl.close()
    return out


def crosscheck_end_windows(pl, reads, matching):
    """All end-window records of phase B for the whole batch, packed-fp16 traced kernel vs packed-int16 one."""
    from porechop_amd.batch import MODE_TRACE

    def records():
        jobs, _ = pl._phase_b_jobs(reads, matching)
        _, out_b, _ = pl._scan_jobs(reads.arena, jobs, MODE_TRACE, pl.p.end_size, with_layout=True)
        pl.aligner.sync()
        return out_b

    a = records()
    ops_a = pl.aligner.trace_ops_per_2_cells()
    pl.aligner.set_int16_only(True)
    try:
        b = records()
        ops_b = pl.aligner.trace_ops_per_2_cells()
    finally:
        pl.aligner.set_int16_only(False)
    return {"device_crosscheck_pairs": int(a.shape[0]), "records_differing": int((a != b).any(dim=1).sum().item()),
            "traced_kernel_ops_per_2_cells": [ops_a, ops_b],
            "what": "every 8-int end-window record of phase B of the whole batch: packed-fp16 traced kernel vs packed-int16 one"}



def step_configs4(pl, reads, n_check, opts, prefilter=False, prune_b=False):
    """BASELINE configs[4] per GPU (-b DIR, middle scan on): phase A (+ MAX all-reduce of the presence table), the
    barcode-kit choice (porechop.py:330-371), the full-barcode rule (porechop.py:410-436), phase B with the barcode
    identities + determine_barcode for every read, phase C over every matching set's start / end sequences
    (porechop.py:541-548, nanopore_read.py:210-243)."""
    matching, orientation, names, bins = configs4_sets(pl, reads, n_check)
    st, et, calls, hits = configs4_scan(pl, reads, matching, bins, opts, prefilter=prefilter, prune_b=prune_b)
    return matching, orientation, names, st, et, calls, hits


def configs4_sets(pl, reads, n_check):
    """The once-per-run part of step_configs4: phase A on the check reads (+ the presence table's MAX all-reduce), the kit
    choice, the full-barcode rule -> (matching set indices, orientation, bin names, bins)."""
    from porechop_amd import panel as rules
    from porechop_amd.distributed import reduce_presence
    from porechop_amd.runner import barcode_bins
    bs, be = pl.phase_a(reads, torch.arange(min(n_check, reads.n), device=reads.off.device))
    bs, be = reduce_presence(bs, be)
    matching = pl.matching_sets(bs, be)
    bsh, beh = bs.cpu().numpy(), be.cpu().numpy()
    index_of = {id(s): i for i, s in enumerate(pl.sets)}
    msets = [pl.sets[i] for i in matching]
    orientation = rules.choose_barcoding_kit(msets, lambda s: bsh[index_of[id(s)]], lambda s: beh[index_of[id(s)]])
    full = rules.add_full_barcode_sets(pl.sets[:pl.n_panel], msets)[len(msets):]
    if full:                                               # native / rapid kits only: none for the PCR barcodes planted here
        known = {s.name: i for i, s in enumerate(pl.sets)}
        new = [s for s in full if s.name not in known]
        if new:
            pl.add_sets(new)
            known = {s.name: i for i, s in enumerate(pl.sets)}
        matching = matching + [known[s.name] for s in full]
    bc_sets = [i for i in matching if rules.is_barcode(pl.sets[i]) and rules.barcode_direction(pl.sets[i]) == orientation]
    names, bins = barcode_bins(pl, bc_sets)
    return matching, orientation, names, bins


def configs4_scan(pl, reads, matching, bins, opts, prefilter=False, prune_b=False):
    """The per-read part: phase B with the barcode identities + determine_barcode, phase C over every matching set's sequences."""
    st, et, calls = pl.phase_b_demux(reads, matching, bins, opts.barcode_threshold, opts.barcode_diff, opts.require_two_barcodes,
                                     prune=prune_b)
    hits = pl.phase_c(reads, st, et, matching, prefilter=prefilter)
    return st, et, calls, hits


def leg_configs4(dev, args, workers, world, rank, barrier):
    """The per-GPU shape of BASELINE configs[4] ("10M reads, full panel + middle scan, read-sharded across 8 GPUs" =
    1.25 M reads per GPU): every rank runs it on its own shard (seed 4 + 1000 * rank), no data-path collective but
    the presence table's MAX all-reduce; the value is all ranks' reads over the slowest rank's time."""
    from dataclasses import asdict
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.runner import Options
    from porechop_amd.synth import make_reads
    from tests.cpu_worker import run_chunk_demux_middle
    p = ScanParams()
    opts = Options()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    pl.n_panel = len(pl.sets)
    import ctypes
    jc0, jd0 = ctypes.c_int64(), ctypes.c_int64()
    pl.aligner.lib.pc_jit_stats(ctypes.byref(jc0), ctypes.byref(jd0))          # process-wide counters: this leg's share below
    n = args.reads4
    fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
    reads = make_reads(n, args.read_len, seed=4 + 1000 * rank, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev,
                       barcodes_start=[a["start"][1] for a in fw], barcodes_end=[a["end"][1] for a in fw])
    n_check = p.check_reads // world + (1 if rank < p.check_reads % world else 0)

    def run(steps, prefilter, prune_b=False):
        out = None
        for _ in range(steps):
            out = step_configs4(pl, reads, n_check, opts, prefilter=prefilter, prune_b=prune_b)
        pl.aligner.sync()
        return out

    def timed_region(steps, prefilter, prune_b=False):
        barrier()
        t0 = time.perf_counter()
        out = run(steps, prefilter, prune_b)
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return out, float(dt.item())

    steps = max(1, min(args.steps, 2))
    run(1, False)                                           # warm-up (kernels come from the library's kernel cache)
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    (matching, orientation, names, st, et, calls, hits), dt = timed_region(steps, False)
    timing = pl.aligner.get_timing()
    pl.aligner.set_timing(False)
    psteps = max(1, min(args.steps, 3))
    run(1, True, True)
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    (_, _, _, st_p, et_p, calls_p, hits_p), dt_p = timed_region(psteps, True, True)
    timing_p = pl.aligner.get_timing()
    pl.aligner.set_timing(False)
    same = bool(hits_p.read.numel() == hits.read.numel() and torch.equal(hits_p.read, hits.read) and
                torch.equal(hits_p.adapter, hits.adapter) and torch.equal(hits_p.start, hits.start) and
                torch.equal(hits_p.end, hits.end) and torch.equal(st_p, st) and torch.equal(et_p, et) and
                np.array_equal(calls_p, calls))
    if world > 1:                                            # EVERY rank's fast variant must equal its own full computation
        flag = torch.tensor([1 if same else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item())
    if rank != 0:
        pl.close()
        return None
    ads = pl.middle_adapter_list(matching)
    A = len(ads)
    mean_trim_len = float((reads.length.to(torch.float64) - st.to(torch.float64) - et.to(torch.float64)).mean().item())
    mean_m = float(np.mean([len(a[1]) for a in ads]))
    truth = reads.truth_barcode.cpu().numpy()
    want = np.array([names.index("BC%02d" % (b + 1)) if "BC%02d" % (b + 1) in names else -2 for b in range(len(fw))])[truth]
    total = n * world
    jc, jd = ctypes.c_int64(), ctypes.c_int64()
    pl.aligner.lib.pc_jit_stats(ctypes.byref(jc), ctypes.byref(jd))
    jc.value -= jc0.value
    jd.value -= jd0.value
    out = {"workload": "BASELINE configs[4] per GPU: %d synthetic %d-bp reads per GPU x %d GPU(s), barcode b ~ U{1..96} at both ends, "
                       "%.0f%% chimeric junctions, full 119-set panel, -b style run: phases A + kit choice + full-barcode rule + B "
                       "(%d pairs per read) + barcode calls + C over all %d sequences of the %d matching sets"
                       % (n, args.read_len, world, args.chimera * 100, sum((pl.sets[i].start is not None) + (pl.sets[i].end is not None) for i in matching),
                          A, len(matching)),
           "n_gpus": world, "reads_per_gpu": n, "steps": steps,
           "reads_per_s": total * steps / dt, "ms_per_step": dt / steps * 1e3, "read_bp_per_s": total * steps / dt * args.read_len,
           "matching_sets": len(matching), "middle_adapters": A, "barcode_orientation": orientation,
           "middle_hits_per_step": int(hits.read.numel()), "mask_rounds": hits.rounds,
           "reads_binned_to_their_planted_barcode": float((calls == want).mean()),
           "kernel_ms_per_step": {k: v[0] / steps for k, v in timing.items()},
           "specialised_kernels": {"compiled_in_this_process": int(jc.value), "loaded_from_the_kernel_cache": int(jd.value)},
           "exact_prefilter": {"reads_per_s": total * psteps / dt_p, "ms_per_step": dt_p / psteps * 1e3, "steps": psteps,
                               "same_trims_calls_and_middle_hits": same, "phase_b_pruned": bool(pl.can_prune_phase_b),
                               "kernel_ms_per_step": {k: v[0] / psteps for k, v in timing_p.items()},
                               "speedup": (dt / steps) / (dt_p / psteps)}}
    jit = timing["score_spec"][1] > 0
    ms, launches, pairs = timing["score_spec"] if jit else timing["score"]
    if launches > 0:
        per_launch_s = ms / 1e3 / launches
        ppl = pairs / launches
        alg = ppl * (mean_trim_len / A + 28.0)
        cells_s = ppl * mean_trim_len * mean_m / per_launch_s
        ops = 5 if jit else 9
        peak = VALU_WAVE_INSTR_PER_S * 64 * 2 / ops / 1e9
        out["roofline"] = {"bound": "valu", "kernel": "pc_spec_score" if jit else "scan_kernel<R,PAD,false>",
                           "achieved": alg / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg / per_launch_s / 1e9 / HBM_PEAK_GBS,
                           "traffic": profile_traffic("configs4", "pc_spec_score" if jit else "scan_kernel<score>", launches / steps),
                           "launches": int(launches), "avg_launch_ms": per_launch_s * 1e3, "algorithmic_bytes_per_launch": alg,
                           "valu": {"achieved_gcups": cells_s / 1e9, "peak_gcups": peak, "frac": cells_s / 1e9 / peak, "ops_per_2_cells": ops},
                           "note": "score-only whole-read scan, %d adapter pairs per read; HBM fraction on ALGORITHMIC bytes "
                                   "(|H|/A + 28 per pair); VALU-bound by construction" % ((A + 1) // 2)}
    out["exact_prefilter"]["roofline"] = prefilter_roofline(timing_p, mean_trim_len, A, leg="configs4_prefilter")
    if args.cpu_seconds > 0 and world == 1:
        seqs, ln = host_seqs(reads, min(n, 2048))
        sets = [(s.name, s.start, s.end) for s in pl.sets]
        mk = lambda c: (c, sets, matching, asdict(p), True, orientation, opts.barcode_threshold, opts.barcode_diff,
                        opts.require_two_barcodes)
        done, dtc, res = cpu_sample(run_chunk_demux_middle, mk, seqs, max(args.cpu_seconds, 12.0), workers, probe=2)
        got = {}
        for r, a_, s_, e_ in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
            if r < done:
                got.setdefault(r, []).append((a_, s_, e_))
        stl, etl = st[:done].cpu().tolist(), et[:done].cpu().tolist()
        cl = [names[k] if k >= 0 else "none" for k in calls[:done]]
        bad = [r for r in range(done) if (stl[r], etl[r], cl[r], got.get(r, [])) != (res[r][0], res[r][1], res[r][2], list(res[r][3]))]
        out["cpu_baseline"] = {"value": done / dtc, "unit": "reads/s", "cores": workers, "kind": baseline_kind(),
                               "sample": "%d reads, phase B + barcode call + phase C (%d adapters), %d worker processes over the %s, %.1f s wall"
                                         % (done, A, workers, kind_text(baseline_kind()), dtc)}
        out["parity"] = {"checked": done, "mismatches": len(bad),
                         "what": "start trim, end trim, barcode call, middle hits (adapter, start, end) per read",
                         "first_mismatching_reads": bad[:8]}
        out["speedup_vs_cpu_baseline"] = out["reads_per_s"] / out["cpu_baseline"]["value"]
    pl.close()
    return out


def _host_memory_available():
    """Bytes this process may still take from the host: the smaller of MemAvailable and the cgroup's headroom."""
    avail = 1 << 62
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            lim = f.read().strip()
        with open("/sys/fs/cgroup/memory.current") as f:
            cur = int(f.read().strip())
        if lim != "max":
            avail = min(avail, int(lim) - cur)
    except Exception:
        pass
    return max(0, avail)


def leg_fixed_total(dev, args, workers, world, rank, barrier):
    """BASELINE configs[4] ITSELF -- a FIXED total of reads (10 M by default) split over the ranks, full 119-set panel,
    barcode calls and the middle scan over every matching set's sequences -- fed from HOST memory: every rank packs its
    chunks to 2 bits per base on its share of the host cores (pc_io_set_thread_limit), uploads them one chunk ahead of the
    scan, unpacks on the device and runs the fast step (exact prefilter + pruned phase B).  This is the STRONG-scaling
    shape: what N GPUs share -- host cores for packing, host memory bandwidth, PCIe -- is inside the timed region.
    Phase A and the set-level rules run once, on every rank's share of the check reads, with the presence table's MAX
    all-reduce (the one collective of the path, porechop.py:286-327)."""
    import threading
    from porechop_amd._lib import load_library
    from porechop_amd.io import pack_reads
    from porechop_amd.pipeline import DeviceReads, Pipeline, ScanParams
    from porechop_amd.runner import Options
    from porechop_amd.synth import make_reads
    p, opts = ScanParams(), Options()
    L = args.read_len
    total = int(args.reads4_total)
    # keep the pinned host copy within what the machine has (a quarter of the available memory over the local ranks)
    budget = _host_memory_available() // (4 * max(1, world))
    reduced = None
    if total // world * L > budget:
        reduced = total = max(world * 1000, int(budget // L) * world)
    n_rank = total // world + (1 if rank < total % world else 0)
    chunk = max(1, min(args.reads4, n_rank))
    bounds = [(a, min(n_rank, a + chunk)) for a in range(0, n_rank, chunk)]
    threads = max(1, workers // world)
    lib = load_library()
    lib.pc_io_set_thread_limit(threads)
    pl = Pipeline(load_panel_sets(), p, device=dev)
    pl.n_panel = len(pl.sets)
    fw = [a for a in load_panel_json() if a["name"].startswith("Barcode ") and "(forward)" in a["name"]]
    h_arena = []
    for k, (a, b) in enumerate(bounds):
        r = make_reads(b - a, L, seed=4 + 1000 * rank + 17 * k, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev,
                       barcodes_start=[x["start"][1] for x in fw], barcodes_end=[x["end"][1] for x in fw])
        h = torch.empty((b - a) * L, dtype=torch.uint8, pin_memory=True)
        h.copy_(r.arena[:(b - a) * L])
        h_arena.append(h)
        del r
    torch.cuda.empty_cache()
    cap = chunk * L
    bufs = [torch.empty(cap + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_pk = [torch.empty((cap + 15) // 16 * 4, dtype=torch.uint8, device=dev) for _ in range(2)]
    h_pk = [torch.empty((cap + 15) // 16 * 4, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    off = torch.arange(chunk, dtype=torch.int64, device=dev) * L
    ln = torch.full((chunk,), L, dtype=torch.int32, device=dev)
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    uploaded = [torch.cuda.Event() for _ in range(2)]
    scanned = [torch.cuda.Event() for _ in range(2)]
    n_check = p.check_reads // world + (1 if rank < p.check_reads % world else 0)
    busy = {"pack": 0.0}

    def stage(k):
        """pack chunk k on this rank's host threads, upload the 2-bit plane, unpack it on the device (copy stream)"""
        sl = k & 1
        lib.pc_io_set_thread_limit(threads)
        nb = int(h_arena[k].numel())
        scanned[sl].synchronize()                       # the scan of chunk k-2 is done with this buffer pair
        t0 = time.perf_counter()
        _, exc = pack_reads(h_arena[k].numpy(), nb, out=h_pk[sl].numpy())
        busy["pack"] += time.perf_counter() - t0
        with torch.cuda.stream(copy_stream):
            d_pk[sl][:(nb + 15) // 16 * 4].copy_(h_pk[sl][:(nb + 15) // 16 * 4], non_blocking=True)
            d_exc = torch.from_numpy(exc).to(dev) if exc.size else None
            pl.aligner.unpack_device(d_pk[sl], nb, d_exc, arena=bufs[sl], pad=64, stream=copy_stream.cuda_stream)
            uploaded[sl].record(copy_stream)

    def run(fast=True, only_first=False):
        sets = None
        results = []
        for e in scanned:
            e.record(main)
        stage(0)
        for k, (a, b) in enumerate(bounds[:1] if only_first else bounds):
            th = None
            if not only_first and k + 1 < len(bounds):
                th = threading.Thread(target=stage, args=(k + 1,))
                th.start()
            main.wait_event(uploaded[k & 1])
            batch = DeviceReads(bufs[k & 1], off[:b - a], ln[:b - a])
            if sets is None:
                sets = configs4_sets(pl, batch, n_check)
            st, et, calls, hits = configs4_scan(pl, batch, sets[0], sets[3], opts, prefilter=fast, prune_b=fast)
            results.append((st.clone(), et.clone(), calls, hits))
            scanned[k & 1].record(main)
            if th is not None:
                th.join()
        pl.aligner.sync()
        return sets, results

    run(True, only_first=True)                           # warm-up: kernels from the cache, scratch buffers sized
    busy["pack"] = 0.0
    barrier()
    t0 = time.perf_counter()
    sets, res_fast = run(True)
    barrier()
    mine = time.perf_counter() - t0
    tall = torch.zeros(world, dtype=torch.float64, device=dev)
    tall[rank] = mine
    if world > 1:
        dist.all_reduce(tall, op=dist.ReduceOp.SUM)
    rank_s = [float(x) for x in tall.cpu()]
    # the fast step must equal the full computation: first chunk of every rank, outside the timed region
    _, res_full = run(False, only_first=True)
    f, g = res_fast[0], res_full[0]
    same = bool(torch.equal(f[0], g[0]) and torch.equal(f[1], g[1]) and np.array_equal(f[2], g[2]) and
                f[3].read.numel() == g[3].read.numel() and torch.equal(f[3].read, g[3].read) and torch.equal(f[3].start, g[3].start) and
                torch.equal(f[3].end, g[3].end))
    flag = torch.tensor([1 if same else 0, n_check, int(sum(int(r[3].read.numel()) for r in res_fast))], dtype=torch.int64, device=dev)
    flags = [torch.zeros_like(flag) for _ in range(world)]
    if world > 1:
        dist.all_gather(flags, flag)
    else:
        flags = [flag]
    flags = [[int(x) for x in t.cpu()] for t in flags]
    pl.close()
    lib.pc_io_set_thread_limit(0)
    if rank != 0:
        return None
    dt = max(rank_s)
    return {"workload": "BASELINE configs[4], fixed total: %d synthetic %d-bp reads split over %d rank(s) (%d per rank in chunks of <= %d), "
                        "barcodes at both ends, %.0f%% chimeras, full panel, demultiplexing + middle scan, from pinned HOST memory: packed to "
                        "2 bits per base on %d host thread(s) per rank, uploaded one chunk ahead, fast step (exact prefilter + pruned phase B)"
                        % (total, L, world, n_rank, chunk, args.chimera * 100, threads),
            "scaling": "strong", "n_gpus": world, "reads_total": total, "reads_total_reduced_to_fit_host_memory": reduced,
            "reads_per_s": total / dt, "wall_s": dt, "ms_by_rank": [x * 1e3 for x in rank_s], "host_threads_per_rank": threads,
            "pack_busy_s_rank0": busy["pack"], "world_size_seen": dist.get_world_size() if world > 1 else 1,
            "backend": dist.get_backend() if world > 1 else "none", "check_reads_by_rank": [x[1] for x in flags],
            "middle_hits_by_rank": [x[2] for x in flags], "fast_same": all(x[0] == 1 for x in flags),
            "matching_sets": len(sets[0])}


def leg_sharded_file(dev, args, world, rank, barrier):
    """File -> file over the ranks (runner.run_sharded): rank r parses the records that start in its W-th of the FASTQ file's
    bytes, scans them on its GPU and writes its own span of the shared output file; the collectives are the presence table
    (MAX), read counts and output sizes.  The file must equal the single-process run's (md5)."""
    import shutil
    from porechop_amd import runner
    from porechop_amd.synth import make_reads
    n, L = args.reads_e2e, args.read_len
    work = [None]
    if rank == 0:
        base = os.environ.get("PC_BENCH_E2E_DIR") or "/tmp"
        work[0] = os.path.join(base, "porechop_amd_sharded_%d" % os.getpid())
        os.makedirs(work[0], exist_ok=True)
        reads = make_reads(n, L, seed=9, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev)
        write_fastq(reads, n, os.path.join(work[0], "in.fastq"))
        del reads
        torch.cuda.empty_cache()
    if world > 1:
        dist.broadcast_object_list(work, src=0)
    inp, out = os.path.join(work[0], "in.fastq"), os.path.join(work[0], "out.fastq")
    try:
        runs = []
        for _ in range(2):
            barrier()
            t0 = time.perf_counter()
            res = runner.run(inp, output=out, device=dev)
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            runs.append((float(dt.item()), res))
        dt, res = min(runs, key=lambda r: r[0])
        shares = torch.zeros(world, dtype=torch.int64, device=dev)
        shares[rank] = res.local_reads if res.local_reads is not None else res.n_reads
        if world > 1:
            dist.all_reduce(shares, op=dist.ReduceOp.SUM)
        if rank != 0:
            return None
        got = file_md5(out)
        single = runner.run_streamed(inp, os.path.join(work[0], "single.fastq"), None, runner.Options(), device=dev)
        want = file_md5(os.path.join(work[0], "single.fastq")) if single is not None else None
        return {"workload": "file -> file over %d rank(s): a %d-read, %.1f GB plain FASTQ file, every rank parses / scans / writes its own "
                            "byte range (runner.run_sharded)" % (world, n, os.path.getsize(inp) / 1e9),
                "n_gpus": world, "reads_per_s": res.n_reads / dt, "wall_s": dt, "reads_by_rank": [int(x) for x in shares.cpu()],
                "stage_seconds": {k: round(v, 3) for k, v in res.seconds.items()}, "md5_equal": bool(want is not None and got == want)}
    finally:
        barrier()
        if rank == 0:
            shutil.rmtree(work[0], ignore_errors=True)


def prefilter_roofline(timing, mean_len, A, leg="prefilter"):
    """Roofline object of the exact prefilter.  Its dominant kernel is the seed scan (seed_scan_kernel): every read byte
    streamed ONCE, whatever the number of adapters, 2 + 4 VALU operations per byte and seed length -- bound by HBM.
    Algorithmic bytes per launch = the windows' bytes; the duration is that kernel's own timed launches (kind
    'seed_scan'), small mask-round launches included in the average.  When no adapter can be seeded the exhaustive
    Myers kernel runs instead (VALU-bound, 12.5 operations per column and adapter piece): then the whole stage is the unit."""
    sms, slaunches, swindows = timing["seed_scan"]
    pms, planches, ppairs = timing["prefilter"]
    if slaunches > 0:
        per_launch_s = sms / 1e3 / slaunches
        alg = swindows / slaunches * mean_len
        return {"bound": "hbm", "kernel": "seed_scan_kernel<NQ> (exact q-gram seeds of all adapters, one pass over the reads)",
                "achieved": alg / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / per_launch_s / 1e9 / HBM_PEAK_GBS,
                "traffic": profile_traffic(leg, "seed_scan_kernel"), "launches": int(slaunches), "avg_launch_ms": per_launch_s * 1e3,
                "algorithmic_bytes_per_launch": alg,
                "whole_stage_ms_per_call": pms / max(1, planches),
                "note": "the stage also runs seed_verify_kernel over the finds (and the exhaustive Myers kernel for adapters without "
                        "seeds); whole_stage_ms_per_call includes them and the stage's one host round trip"}
    if planches <= 0:
        return None
    per_launch_s = pms / 1e3 / planches
    alg = ppairs / planches / max(A, 1) * mean_len
    steps_s = alg * A / per_launch_s
    peak_steps = 1024 * 2.4e9 / 2.0 * 64 / 12.5            # plain 32-bit VALU ops: one wave64 instruction per 2 cycles per SIMD
    return {"bound": "valu", "kernel": "prefilter_kernel<P> (Myers bit-vector edit distance, one lane per read chunk, P adapter pieces per lane)",
            "achieved": alg / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / per_launch_s / 1e9 / HBM_PEAK_GBS,
            "traffic": profile_traffic(leg, "prefilter_kernel"), "launches": int(planches), "avg_launch_ms": per_launch_s * 1e3,
            "algorithmic_bytes_per_launch": alg,
            "valu": {"achieved_column_updates_per_s": steps_s, "peak_column_updates_per_s": peak_steps, "frac": steps_s / peak_steps,
                     "ops_per_column_and_adapter": 12.5}}


def library_fingerprint():
    """sha1 of the loaded libporechop_amd.so (what a profiles/*_summary.json must have been taken with)."""
    import hashlib
    import porechop_amd
    h = hashlib.sha1()
    with open(porechop_amd.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def device_sources_fingerprint():
    """tools/device_fingerprint.py over the working tree: sha1 of every source that decides what the GPU does."""
    try:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import device_fingerprint
        return device_fingerprint.fingerprint()
    except Exception:
        return None


_PROFILE = None


def profile_traffic(leg, kernel, launches_per_step=None):
    """HBM-side traffic (FETCH_SIZE + WRITE_SIZE, bytes per launch) of the kernel family `kernel` in `leg`, from the newest
    profiles/*_summary.json (tools/profile_round3.sh + summarize_profile3.py: rocprofv3 --pmc, one counter per pass, a
    process that runs only that leg) -- ONLY if that summary was taken with the very library now loaded (its recorded sha1
    equals the loaded .so's, or its device-source fingerprint equals the working tree's): a kernel change without a re-profile
    reports null, never stale counters.  The summary holds
    per-step sums; `launches_per_step` is this run's count of timed launches of the kernel per step (None: the profile's own)."""
    global _PROFILE
    if _PROFILE is None:
        _PROFILE = {}
        try:
            import glob
            summ = sorted(glob.glob(os.path.join(REPO, "profiles", "*_summary.json")))
            if summ:
                with open(summ[-1]) as f:
                    sj = json.load(f)
                # ... or with a library built from the same kernels, launch planning and flags: only the host I/O code
                # (FASTQ / gzip in and out) differs (tools/device_fingerprint.py)
                if sj.get("library_sha1") == library_fingerprint() or \
                        (sj.get("device_sources_sha1") and sj.get("device_sources_sha1") == device_sources_fingerprint()):
                    _PROFILE = sj
                    _PROFILE["_path"] = os.path.relpath(summ[-1], REPO)
        except Exception:
            _PROFILE = {}
    try:
        kk = _PROFILE.get("legs", {}).get(leg, {}).get("kernels", {}).get(kernel)
        if kk and "FETCH_SIZE_bytes_per_step" in kk and "WRITE_SIZE_bytes_per_step" in kk:
            per_step = kk["FETCH_SIZE_bytes_per_step"] + kk["WRITE_SIZE_bytes_per_step"]
            return per_step / max(1e-9, launches_per_step if launches_per_step else kk["launches_per_step"])
    except Exception:
        pass
    return None


def live_traffic(kernel_substring, leg="headline", steps=2, timeout_s=300):
    """FETCH_SIZE + WRITE_SIZE of one kernel family measured NOW: `rocprofv3 --pmc <counter>` (each counter in a pass of its own,
    no trace domain beside it: MI355X_MICROARCH.md's recipe) over a child that runs `steps` steps of `leg` and nothing else
    (tools/run_leg.py; its first step is warm-up like every launch after it: the counters are per launch and do not care).
    -> {"bytes_per_launch", "FETCH_SIZE", "WRITE_SIZE", "launches", "seconds"} or None (no rocprofv3, a failed pass: the line then
    keeps the profile's figure).  KB as reported -> bytes; no x2 (see the comment at the call)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None                    # this run is itself being profiled: no profiler inside a profiler
    t0 = time.perf_counter()
    out = {}
    launches = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        work = tempfile.mkdtemp(prefix="pc_live_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", work, "-o", "pmc", "--", sys.executable,
                                os.path.join(REPO, "tools", "run_leg.py"), leg, str(steps)], cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            total, disp = 0.0, set()
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if kernel_substring in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        total += float(row["Counter_Value"]) * 1024.0
                        disp.add(row["Dispatch_Id"])
            if not disp:
                return None
            out[counter] = total / len(disp)
            launches = len(disp)
        except Exception:
            return None
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return {"bytes_per_launch": out["FETCH_SIZE"] + out["WRITE_SIZE"], "FETCH_SIZE": out["FETCH_SIZE"], "WRITE_SIZE": out["WRITE_SIZE"],
            "launches": launches, "seconds": time.perf_counter() - t0}


def leg_host_buffers(dev, args):
    """The headline workload with the reads starting in (pinned) HOST memory: every step uploads all reads over PCIe again,
    in batches on a copy stream into two device buffers while the previous batch is being scanned on the compute stream
    (phase A runs on the first batch, which holds the check reads).  Twice: the reads held at 2 BITS per base, the form
    pc_pack_reads gives them once at ingest (north_star: "2-bit-packed read windows"; pc_unpack_device turns a batch into
    the byte arena on the GPU) -- `reads_per_s` -- and at one byte per base as earlier rounds did (`bytes_per_base_1`).
    This is the PCIe-inclusive rate of DESIGN.md section 6 -- never `value`, which is measured HBM-resident."""
    from porechop_amd.io import pack_reads
    from porechop_amd.pipeline import Pipeline, ScanParams, DeviceReads
    from porechop_amd.synth import make_reads
    p = ScanParams()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    n, nb = args.reads, int(os.environ.get("PC_BENCH_H2D_BATCHES", "8"))
    reads = make_reads(n, args.read_len, seed=3, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev)
    total = int(reads.arena.shape[0])
    h_arena = torch.empty(total, dtype=torch.uint8, pin_memory=True)
    h_arena.copy_(reads.arena)
    h_off, h_len = reads.off.cpu().pin_memory(), reads.length.cpu().pin_memory()
    del reads
    torch.cuda.empty_cache()
    # Batches: at one byte per base the upload (140 ms) bounds the step and eight batches hide all but the first behind scans;
    # at 2 bits per base the scan bounds it, and fewer, larger batches keep its launches at full size (a 125 k-read batch
    # runs phases B and C at ~0.7 of the rate of a 333 k-read one: launch tails, one host round trip per mask round).
    nb_packed = int(os.environ.get("PC_BENCH_H2D_BATCHES_PACKED", "2"))

    def layout(k_batches):
        per_ = (n + k_batches - 1) // k_batches
        bounds_ = [(i, min(n, i + per_)) for i in range(0, n, per_)]
        first_ = [int(h_off[a]) for a, _ in bounds_] + [int(h_off[n - 1]) + int(h_len[n - 1])]
        return per_, bounds_, first_
    lay = {False: layout(nb), True: layout(nb_packed)}
    per = max(lay[False][0], lay[True][0])
    cap = (max(f[k + 1] - f[k] for _, b, f in lay.values() for k in range(len(b))) + 64 + 15) // 16 * 16
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(2)]
    offs = [torch.empty(per, dtype=torch.int64, device=dev) for _ in range(2)]
    lens = [torch.empty(per, dtype=torch.int32, device=dev) for _ in range(2)]
    # the 2-bit form of every batch, packed once (ingest-time work, timed and reported, not part of a step)
    t0 = time.perf_counter()
    h_pk, h_exc = [], []
    _, bounds, first = lay[True]
    for k in range(len(bounds)):
        nbases = first[k + 1] - first[k]
        out = torch.empty((nbases + 15) // 16 * 4, dtype=torch.uint8, pin_memory=True)
        _, exc = pack_reads(h_arena.numpy()[first[k]:first[k + 1]], nbases, out=out.numpy())
        h_pk.append(out)
        h_exc.append(torch.from_numpy(exc).pin_memory() if exc.size else None)
    pack_s = time.perf_counter() - t0
    d_pk = [torch.zeros(max(int(x.numel()) for x in h_pk) + 64, dtype=torch.uint8, device=dev) for _ in range(2)]   # (+64: the packed scan fetches whole 16-byte blocks)
    d_exc = [torch.empty(max([1] + [int(x.numel()) for x in h_exc if x is not None]), dtype=torch.int64, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    uploaded = [torch.cuda.Event() for _ in range(2)]
    scanned = [torch.cuda.Event() for _ in range(2)]
    packed, prefilter, lazy = [True], [False], [False]      # lazy: the uploaded plane is NOT unpacked (DeviceReads.packed_only)
    # A continuous stream of batches: uploads run ONE BATCH AHEAD of the scans, across step boundaries too (the first batch
    # of step s+1 crosses the link under the last scan of step s); buffers alternate by a global batch counter.
    seq = {"next_upload": 0, "next_scan": 0, "primed": False}

    def upload(k):
        _, bounds, first = lay[packed[0]]
        a, b = bounds[k]
        s = seq["next_upload"] & 1
        seq["next_upload"] += 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(scanned[s])                      # the scan of batch k-2 is done with this buffer
            nbytes = first[k + 1] - first[k]
            if packed[0]:
                d_pk[s][:h_pk[k].numel()].copy_(h_pk[k], non_blocking=True)
                ne = 0 if h_exc[k] is None else int(h_exc[k].numel())
                if ne:
                    d_exc[s][:ne].copy_(h_exc[k], non_blocking=True)
                if not lazy[0]:
                    pl.aligner.unpack_device(d_pk[s], nbytes, d_exc[s][:ne] if ne else None, arena=bufs[s], pad=64,
                                             stream=copy_stream.cuda_stream)
            else:
                bufs[s][:nbytes].copy_(h_arena[first[k]:first[k + 1]], non_blocking=True)
                bufs[s][nbytes:nbytes + 64].fill_(ord("N"))
            offs[s][:b - a].copy_(h_off[a:b], non_blocking=True)
            lens[s][:b - a].copy_(h_len[a:b], non_blocking=True)
            uploaded[s].record(copy_stream)

    def step():
        _, bounds, first = lay[packed[0]]
        matching, hits_n = None, 0
        if not seq["primed"]:
            upload(0)
            seq["primed"] = True
        for k, (a, b) in enumerate(bounds):
            s = seq["next_scan"] & 1
            seq["next_scan"] += 1
            upload((k + 1) % len(bounds))                           # in flight while batch k is scanned (k = last: the next step's first)
            main.wait_event(uploaded[s])
            if lazy[0]:
                ne = 0 if h_exc[k] is None else int(h_exc[k].numel())
                batch = DeviceReads.packed_only(pl.aligner, d_pk[s], first[k + 1] - first[k], d_exc[s][:ne] if ne else None,
                                                (offs[s][:b - a] - first[k]).contiguous(), lens[s][:b - a], end_size=p.end_size)
            else:
                batch = DeviceReads(bufs[s], offs[s][:b - a] - first[k], lens[s][:b - a])
            if k == 0:
                bs, be = pl.phase_a(batch, torch.arange(min(p.check_reads, b - a), device=dev))
                matching = pl.matching_sets(bs, be)
            st, et = pl.phase_b(batch, matching)
            hits = pl.phase_c(batch, st, et, matching, prefilter=prefilter[0])
            hits_n += int(hits.read.numel())
            scanned[s].record(main)
        return matching, hits_n

    def sync():
        pl.aligner.sync()
        torch.cuda.synchronize()

    def upload_alone():
        _, bounds, first = lay[packed[0]]
        sync()
        t0 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            for k in range(len(bounds)):
                if packed[0]:
                    d_pk[k & 1][:h_pk[k].numel()].copy_(h_pk[k], non_blocking=True)
                    pl.aligner.unpack_device(d_pk[k & 1], first[k + 1] - first[k], None, arena=bufs[k & 1], pad=64,
                                             stream=copy_stream.cuda_stream)
                else:
                    bufs[k & 1][:first[k + 1] - first[k]].copy_(h_arena[first[k]:first[k + 1]], non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for e in scanned:
        e.record(main)
    steps = max(1, min(args.steps, 5))
    res = {}
    def restart():
        sync()
        seq.update(next_upload=0, next_scan=0, primed=False)
        for e in scanned:
            e.record(main)
        sync()

    for mode in (False, True):
        packed[0] = mode
        restart()
        (matching, hits_n), dt = timed(step, steps, max(1, min(args.warmup, 2)), sync)
        dt_up = upload_alone()
        sent = sum(int(x.numel()) for x in h_pk) if mode else total
        res[mode] = {"reads_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3, "h2d_ms_alone": dt_up * 1e3,
                     "h2d_gb_per_s_alone": sent / dt_up / 1e9, "bytes_uploaded_per_step": sent, "middle_hits_per_step": hits_n,
                     "batches": len(lay[mode][1])}
    out = {"workload": "BASELINE configs[3] from pinned host memory: %d reads x %d bp uploaded every step in %d batches at 2 bits per "
                       "base (pc_pack_reads once at ingest; pc_unpack_device per batch), uploads one batch ahead of the scans, across "
                       "step boundaries too" % (n, args.read_len, len(lay[True][1])),
           "packed": True, "steps": steps, "pack_once_s": pack_s, "pack_gb_per_s": total / pack_s / 1e9,
           "exceptions": sum(0 if x is None else int(x.numel()) for x in h_exc),
           "same_hits_both_forms": res[True]["middle_hits_per_step"] == res[False]["middle_hits_per_step"],
           "bytes_per_base_1": res[False], "matching_sets": [pl.sets[i].name for i in matching]}
    out.update(res[True])
    # the same, the middle scan behind the exact prefilter (config.exact_prefilter's step, fed from the host)
    prefilter[0] = True
    restart()
    (_, hits_f), dt_f = timed(step, steps, 1, sync)
    out["exact_prefilter"] = {"reads_per_s": n * steps / dt_f, "ms_per_step": dt_f / steps * 1e3,
                              "same_hits": hits_f == res[True]["middle_hits_per_step"]}
    # ... and with the uploaded plane left PACKED: no pc_unpack_device pass, the prefilter scans the plane, only end windows and
    # survivors become bytes.  (The link bounds this leg: 2 bits per base of 8 Gbase at ~55 GB/s is 36 ms = 27.5 M reads/s.)
    try:
        lazy[0] = True
        restart()
        (_, hits_l), dt_l = timed(step, steps, 1, sync)
        out["exact_prefilter"]["kept_packed"] = {"reads_per_s": n * steps / dt_l, "ms_per_step": dt_l / steps * 1e3,
                                                 "same_hits": hits_l == res[True]["middle_hits_per_step"]}
    except Exception as e:
        out["exact_prefilter"]["kept_packed"] = {"failed": repr(e)}
    lazy[0] = False
    pl.close()
    return out



def leg_end_to_end(dev, args):
    """FASTQ file in -> trimmed / split FASTQ file out, through porechop_amd.runner (SURVEY.md 8f-1..3 around the hot
    path): the configs[3] read set written as a plain FASTQ file, runner.run() on it (the streamed path: a loader thread
    parses block k+1 with all host cores while block k is scanned and block k-1 is formatted and written), wall clock
    of the whole call.  Files live on tmpfs when /dev/shm has the room (stated in the result), else under /tmp (page
    cache).  The streamed run's output must equal the whole-file path's byte for byte."""
    import filecmp
    import shutil
    from porechop_amd import runner
    from porechop_amd.synth import make_reads
    n, L = args.reads_e2e, args.read_len
    need = n * (2 * L + 16) * 2.2
    base = os.environ.get("PC_BENCH_E2E_DIR")
    if not base:          # page cache first: on the GPU box one file takes 15 GB/s there and 9 GB/s on tmpfs (tools/ubench_write.cpp)
        try:
            base = "/tmp" if shutil.disk_usage("/tmp").free > need else "/dev/shm"
        except Exception:
            base = "/tmp"
    work = os.path.join(base, "porechop_amd_e2e_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)
    try:
        reads = make_reads(n, L, seed=9, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev)
        inp = os.path.join(work, "in.fastq")
        in_bytes = write_fastq(reads, n, inp)
        del reads
        torch.cuda.empty_cache()
        runs = []
        out_s = os.path.join(work, "out_streamed.fastq")
        for _ in range(3):
            if os.path.exists(out_s):
                os.remove(out_s)                       # (giving a 6.4 GB file's pages back is not part of the next run)
            t0 = time.perf_counter()
            res = runner.run(inp, output=out_s, device=dev)
            dt = time.perf_counter() - t0
            runs.append({"wall_s": dt, "reads_per_s": res.n_reads / dt, "stage_seconds": {k: round(v, 3) for k, v in res.seconds.items()}})
        out_w = os.path.join(work, "out_whole.fastq")
        old = os.environ.get("PC_STREAM_BLOCK_BYTES")
        os.environ["PC_STREAM_BLOCK_BYTES"] = str(1 << 50)          # whole-file path
        try:
            t0 = time.perf_counter()
            res_w = runner.run(inp, output=out_w, device=dev)
            dt_w = time.perf_counter() - t0
        finally:
            if old is None:
                del os.environ["PC_STREAM_BLOCK_BYTES"]
            else:
                os.environ["PC_STREAM_BLOCK_BYTES"] = old
        same = filecmp.cmp(out_s, out_w, shallow=False)
        # the MEDIAN of the three runs is the leg's number (the first run pays the page cache's first touch, the best one is luck)
        best = sorted(runs, key=lambda r: r["reads_per_s"])[len(runs) // 2]
        return {"workload": "end to end: %d synthetic %d-bp reads (configs[3] shape) as a %.1f GB plain FASTQ file -> trimmed / split FASTQ "
                            "(%.1f GB) through porechop_amd.runner.run (streamed: ingest, scan and writing of successive 256 MB blocks overlap)"
                            % (n, L, in_bytes / 1e9, os.path.getsize(out_s) / 1e9),
                "files_on": "tmpfs (/dev/shm)" if base.startswith("/dev/shm") else base + " (disk-backed, through the page cache)",
                "reads_per_s": best["reads_per_s"], "wall_s": best["wall_s"], "runs": runs, "reads_per_s_is": "median of the runs",
                "best_reads_per_s": max(r["reads_per_s"] for r in runs),
                "whole_file_path": {"wall_s": dt_w, "reads_per_s": res_w.n_reads / dt_w,
                                    "stage_seconds": {k: round(v, 3) for k, v in res_w.seconds.items()}},
                "streamed_output_identical_to_whole_file_output": bool(same),
                "matching_sets": res.matching_sets, "reads_with_middle_hits": int(res.middle_hit_reads),
                "input_gb_per_s": in_bytes / 1e9 / best["wall_s"]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def leg_end_to_end_gz(dev, args, workers):
    """.fastq.gz in -> .fastq.gz out through porechop_amd.runner (real nanopore input is mostly gzip): the configs[3] read
    set as (a) a file of sized members -- what this library and bgzip write: inflated by all cores -- and (b) ONE pigz-style
    member -- what gzip / pigz write: inflated by a producer thread ahead of the streamed route --, output deflated by all
    cores straight from the formatter threads (porechop.py:640-651,685-729 shells out to `pigz -p <threads>` over a
    temporary file; misc.py:151-168 reads through Python's gzip module).  The gunzip-ed output must be the plain route's
    output, and the unchanged reference CLI runs .gz -> .gz on the first --cli-reads reads of the same file."""
    import hashlib
    import shutil
    import subprocess
    from porechop_amd import io as pio
    from porechop_amd import runner
    from porechop_amd.synth import make_reads
    n, L = args.reads_e2e, args.read_len
    need = n * (2 * L + 16) * 2.6
    base = os.environ.get("PC_BENCH_E2E_DIR")
    if not base:
        try:
            base = "/tmp" if shutil.disk_usage("/tmp").free > need else "/dev/shm"
        except Exception:
            base = "/tmp"
    work = os.path.join(base, "porechop_amd_e2egz_%d" % os.getpid())
    os.makedirs(work, exist_ok=True)

    def gunzip_md5(path):
        h = hashlib.md5()
        with subprocess.Popen(["gzip", "-dc", path], stdout=subprocess.PIPE) as pr:
            for chunk in iter(lambda: pr.stdout.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()
    try:
        reads = make_reads(n, L, seed=9, start_frac=0.9, end_frac=0.5, chimera_frac=args.chimera, device=dev)
        inp = os.path.join(work, "in.fastq")
        in_bytes = write_fastq(reads, n, inp)
        small = os.path.join(work, "small.fastq")
        k_small = max(1, min(args.cli_reads, n))
        write_fastq(reads, k_small, small)
        del reads
        torch.cuda.empty_cache()
        sized, single = os.path.join(work, "in_sized.fastq.gz"), os.path.join(work, "in_single.fastq.gz")
        t0 = time.perf_counter()
        pio.gzip_file(inp, sized)
        dt_deflate = time.perf_counter() - t0
        pio.gzip_file(inp, single, single_member=True)
        pio.gzip_file(small, small + ".gz", single_member=True)
        # (c) `cat part*.fastq.gz`: 64 ordinary members back to back, no size subfields -- found by guessing, inflated ahead by workers
        catted = os.path.join(work, "in_cat.fastq.gz")
        with open(inp, "rb") as f_in, open(catted, "wb") as f_cat:
            rec_bytes = in_bytes // n                                              # (write_fastq: fixed-width records)
            per = max(1, (in_bytes // 64) // rec_bytes) * rec_bytes
            k_ = 0
            while True:
                blob = f_in.read(per if k_ < 63 else in_bytes)
                if not blob:
                    break
                part = os.path.join(work, "part.fastq")
                with open(part, "wb") as f_p:
                    f_p.write(blob)
                pio.gzip_file(part, part + ".gz", single_member=True)
                with open(part + ".gz", "rb") as f_g:
                    shutil.copyfileobj(f_g, f_cat, 1 << 24)
                k_ += 1
            os.remove(part); os.remove(part + ".gz")
        t0 = time.perf_counter()
        rs = pio.ReadSet(sized)                                   # whole-file loader: all cores on the sized members
        dt_inflate = time.perf_counter() - t0
        n_loaded = rs.count
        rs.close()
        out_plain = os.path.join(work, "out.fastq")
        runner.run(inp, output=out_plain, device=dev)
        want = file_md5(out_plain)
        plain_out_bytes = os.path.getsize(out_plain)
        os.remove(out_plain)
        legs = {}
        first_out = None
        for name, src, reps in (("sized_members", sized, 2), ("single_member", single, 1), ("concatenated_members", catted, 1)):
            runs = []
            out_gz = os.path.join(work, "out_%s.fastq.gz" % name)
            for _ in range(reps):
                if os.path.exists(out_gz):
                    os.remove(out_gz)
                t0 = time.perf_counter()
                res = runner.run(src, output=out_gz, device=dev)
                dt = time.perf_counter() - t0
                runs.append({"wall_s": dt, "reads_per_s": res.n_reads / dt, "stage_seconds": {k: round(v, 3) for k, v in res.seconds.items()}})
            best = max(runs, key=lambda r: r["reads_per_s"])
            # (the first layout's output is gunzip-ed and hashed; a later layout's output that equals it byte for byte -- same
            # blocks, same spans, a deterministic compressor -- needs no second 6 GB through `gzip -dc`)
            import filecmp
            if first_out is not None and filecmp.cmp(out_gz, first_out[0], shallow=False):
                md5_ok = first_out[1]
            else:
                md5_ok = bool(gunzip_md5(out_gz) == want)
            legs[name] = {"reads_per_s": best["reads_per_s"], "wall_s": best["wall_s"], "runs": runs,
                          "input_gz_bytes": os.path.getsize(src), "output_gz_bytes": os.path.getsize(out_gz),
                          "gunzipped_output_md5_equals_plain_route": md5_ok}
            if first_out is None:
                first_out = (out_gz, md5_ok)
            else:
                os.remove(out_gz)
        out = {"workload": "end to end, gzip both ways: %d synthetic %d-bp reads (configs[3] shape), %.1f GB of FASTQ as .fastq.gz -> "
                           "trimmed / split .fastq.gz through porechop_amd.runner.run (streamed)" % (n, L, in_bytes / 1e9),
               "files_on": "tmpfs (/dev/shm)" if base.startswith("/dev/shm") else base + " (disk-backed, through the page cache)",
               "reads_per_s": legs["sized_members"]["reads_per_s"], "wall_s": legs["sized_members"]["wall_s"],
               "single_member_reads_per_s": legs["single_member"]["reads_per_s"],
               "concatenated_members_reads_per_s": legs["concatenated_members"]["reads_per_s"],
               "md5_equal": bool(all(v["gunzipped_output_md5_equals_plain_route"] for v in legs.values())),
               "deflate_gb_per_s": in_bytes / 1e9 / dt_deflate, "deflate_cores": workers,
               "deflate_ratio": os.path.getsize(sized) / in_bytes,
               "inflate_sized_members_gb_per_s": in_bytes / 1e9 / dt_inflate, "reads_loaded": n_loaded,
               "plain_output_gb": plain_out_bytes / 1e9, "by_input_layout": legs,
               "deflate": "libdeflate level 3 (dlopen)" if os.path.exists("/lib/x86_64-linux-gnu/libdeflate.so.0") else "zlib level 6"}
        # the unchanged reference CLI, .gz -> .gz, on the first reads of the same file (its compressor: pigz -p <threads> where
        # the box has pigz, else gzip -- porechop.py:644-651), and this runner on that same small file
        from tests.ref_cli import staged
        if staged() and args.cli_reads > 0 and args.cpu_seconds > 0:
            try:
                ref_out = os.path.join(work, "ref_out.fastq.gz")
                rep, wall = run_ref_cli(small + ".gz", ref_out, workers)
                t0 = time.perf_counter()
                runner.run(small + ".gz", output=os.path.join(work, "small_out.fastq.gz"), device=dev)
                dt_small = time.perf_counter() - t0
                out["reference_cli"] = {"reads": k_small, "threads": workers, "reads_per_s": k_small / rep["main_s"], "main_s": rep["main_s"],
                                        "compressor": "pigz -p %d" % workers if shutil.which("pigz") else "gzip (no pigz on this box)",
                                        "md5_equal": bool(gunzip_md5(ref_out) == gunzip_md5(os.path.join(work, "small_out.fastq.gz"))),
                                        "runner_on_the_same_file_reads_per_s": k_small / dt_small}
                out["speedup_vs_reference_cli"] = out["reads_per_s"] / out["reference_cli"]["reads_per_s"]
            except Exception as e:
                out["reference_cli"] = {"failed": repr(e)}
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def leg_ragged(dev, args, workers, uniform_bp_per_s):
    """The headline workload (configs[3] shape: phases A + B + C, 1 % chimeras) on a realistic length
    distribution instead of exactly 8 000 bases per read: log-normal, mean 8 kb, sigma 0.6."""
    from dataclasses import asdict
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_ragged_reads
    from tests.cpu_worker import run_chunk
    p = ScanParams()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    n = args.reads
    reads = make_ragged_reads(n, mean_len=args.read_len, sigma=0.6, min_len=20, seed=5, start_frac=0.9, end_frac=0.5,
                              chimera_frac=args.chimera, device=dev)
    bases = int(reads.length.to(torch.int64).sum().item())

    def sync():
        pl.aligner.sync()
        torch.cuda.synchronize()
    steps = max(1, min(args.steps, 5))
    (matching, st, et, hits), dt = timed(lambda: one_step(pl, reads, p.check_reads, 1), steps, max(1, min(args.warmup, 2)), sync)
    out = {"workload": "configs[3] shape on log-normal read lengths: %d reads, mean %d bp, sigma 0.6 (min %d, max %d bp), %.0f%% chimeras, "
                       "phases A + B + C" % (n, args.read_len, int(reads.length.min()), int(reads.length.max()), args.chimera * 100),
           "reads_per_s": n * steps / dt, "read_bp_per_s": bases * steps / dt, "ms_per_step": dt / steps * 1e3,
           "bp_per_s_vs_uniform_lengths": bases * steps / dt / uniform_bp_per_s,
           "middle_hits_per_step": int(hits.read.numel()), "matching_sets": [pl.sets[i].name for i in matching]}
    if args.cpu_seconds > 0:
        k = min(n, 512)
        host = reads.arena[: int(reads.off[k - 1]) + int(reads.length[k - 1])].cpu().numpy().tobytes().decode("ascii")
        offs, lens = reads.off[:k].cpu().tolist(), reads.length[:k].cpu().tolist()
        seqs = [host[o:o + l] for o, l in zip(offs, lens)]
        sets = [(s.name, s.start, s.end) for s in pl.sets]
        done, dtc, res = cpu_sample(run_chunk, lambda c: (c, sets, matching, asdict(p), True), seqs, 1e9, workers)
        got = {}
        for r, a, s_, e_ in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
            if r < done:
                got.setdefault(r, []).append((a, s_, e_))
        stl, etl = st[:done].cpu().tolist(), et[:done].cpu().tolist()
        bad = [r for r in range(done) if (stl[r], etl[r], got.get(r, [])) != (res[r][0], res[r][1], list(res[r][2]))]
        out["parity"] = {"checked": done, "mismatches": len(bad), "what": "start trim, end trim, middle hits per read",
                         "first_mismatching_reads": bad[:8]}
    pl.close()
    return out


def leg_ultralong(dev, args, workers):
    """Ultra-long reads: the configs[3] shape (phases A + B + C, chimeras) on a log-normal length distribution with a tail
    far past 65 535 bases (mean 20 kb, sigma 1.2: the longest of 40 000 reads is about a megabase) -- the columns beyond
    u16 of the score kernels, the chunked launch plan, the prefilter's chunks and the mask-and-realign copies at those
    lengths.  Parity: the reference's per-read logic over the compiled reference on the LONGEST reads and on the first ones."""
    from dataclasses import asdict
    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_ragged_reads
    from tests.cpu_worker import run_chunk
    p = ScanParams()
    pl = Pipeline(load_panel_sets(), p, device=dev)
    n = max(1000, args.reads // 25)
    reads = make_ragged_reads(n, mean_len=20000, sigma=1.2, min_len=20, seed=9, start_frac=0.9, end_frac=0.5,
                              chimera_frac=5 * args.chimera, device=dev)
    bases = int(reads.length.to(torch.int64).sum().item())
    over = int((reads.length > 65535).sum().item())

    def sync():
        pl.aligner.sync()
        torch.cuda.synchronize()
    steps = max(1, min(args.steps, 3))
    (matching, st, et, hits), dt = timed(lambda: one_step(pl, reads, p.check_reads, 1), steps, 1, sync)
    (_, st_f, et_f, hits_f), dt_f = timed(lambda: one_step(pl, reads, p.check_reads, 1, prefilter=True), steps, 1, sync)
    same = bool(torch.equal(st, st_f) and torch.equal(et, et_f) and hits_f.read.numel() == hits.read.numel() and
                torch.equal(hits_f.read, hits.read) and torch.equal(hits_f.start, hits.start) and torch.equal(hits_f.end, hits.end))
    beyond = int((hits.start > 65535).sum().item())
    out = {"workload": "configs[3] shape on log-normal read lengths with an ultra-long tail: %d reads, mean 20 kb, sigma 1.2 "
                       "(max %d bp, %d reads above 65 535 bp), %.0f%% chimeras, phases A + B + C"
                       % (n, int(reads.length.max()), over, 5 * args.chimera * 100),
           "reads": n, "max_len": int(reads.length.max()), "reads_over_65535": over,
           "reads_per_s": n * steps / dt, "read_bp_per_s": bases * steps / dt, "ms_per_step": dt / steps * 1e3,
           "prefiltered_read_bp_per_s": bases * steps / dt_f, "prefiltered_same": same,
           "middle_hits_per_step": int(hits.read.numel()), "middle_hits_beyond_column_65535": beyond}
    if args.cpu_seconds > 0:
        order = torch.argsort(reads.length, descending=True, stable=True)[:24].cpu().tolist()
        sample = sorted(set(order) | set(range(min(n, 72))))
        seqs = []
        for r in sample:
            o, l = int(reads.off[r]), int(reads.length[r])
            seqs.append(reads.arena[o:o + l].cpu().numpy().tobytes().decode("ascii"))
        sets = [(s.name, s.start, s.end) for s in pl.sets]
        done, dtc, res = cpu_sample(run_chunk, lambda c: (c, sets, matching, asdict(p), True), seqs, 1e9, workers)
        got = {}
        for r, a, s_, e_ in zip(hits.read.cpu().tolist(), hits.adapter.cpu().tolist(), hits.start.cpu().tolist(), hits.end.cpu().tolist()):
            got.setdefault(r, []).append((a, s_, e_))
        stl, etl = st.cpu().tolist(), et.cpu().tolist()
        bad = [r for k, r in enumerate(sample[:done]) if (stl[r], etl[r], got.get(r, [])) != (res[k][0], res[k][1], list(res[k][2]))]
        out["parity"] = {"checked": done, "mismatches": len(bad), "longest_checked_bp": max(len(x) for x in seqs[:done]) if done else 0,
                         "what": "start trim, end trim, middle hits per read: the 24 longest reads and the first 72",
                         "first_mismatching_reads": bad[:8]}
    pl.close()
    return out


def _r(x, sig=6):
    """Numbers of the printed line: 6 significant digits."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except Exception:
        return x


def _pick(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def compact_line(full):
    """The ONE line bench.py prints: every leg's numbers, no prose, under 8 KB (the driver keeps the last 8 KB of stdout and,
    in its parsed copy, only scalars one level deep) -- so every leg's rate, parity count and roofline fraction is ALSO a flat
    scalar in `config`.  What the keys mean is in DESIGN.md section 9; the unabridged record goes to --full-json."""
    cfg, roof, cpu = full.get("config", {}), full.get("roofline") or {}, full.get("cpu_baseline") or {}
    also = cfg.get("also_measured", {})
    legs = {}

    def leg(name, **kw):
        d = {k: _r(v) for k, v in kw.items() if v is not None}
        # counter bytes / algorithmic bytes of the leg's dominant kernel (profiles/<round>_summary.json): the wasted re-reads at a glance
        for t, a in (("traffic", "alg_bytes_per_launch"), ("seed_scan_traffic", "seed_scan_alg_bytes")):
            if d.get(t) and d.get(a):
                d["traffic_ratio"] = _r(d[t] / d[a], 4)
        if d.get("valu_frac") is not None:
            # against the guide's nominal 2-cycle wave64 issue (MI355X_MICROARCH.md: 1.2 G wave-instr/s/SIMD at 2.4 GHz) -- twice the
            # 4-cycle ceiling valu_frac is priced against (the rate every packed-16 op measures chip-wide, tools/ubench_valu.hip)
            d["valu_frac_nominal"] = _r(d["valu_frac"] / 2.0)
        legs[name] = d

    c1 = also.get("configs1", {})
    if c1:
        leg("configs1", failed=c1.get("failed"), reads_per_s=c1.get("reads_per_s"), ms_per_step=c1.get("ms_per_step"),
            phase_b_reads_per_s=_pick(c1, "phase_b", "reads_per_s"), parity_checked=_pick(c1, "parity", "checked"),
            mismatches=_pick(c1, "parity", "mismatches"), roofline_frac=_pick(c1, "roofline", "frac"),
            valu_frac=_pick(c1, "roofline", "valu", "frac"), avg_launch_ms=_pick(c1, "roofline", "avg_launch_ms"),
            phase_b_gcups=_pick(c1, "roofline", "valu", "achieved_gcups"),
            step_valu_frac=_pick(c1, "step_roofline", "valu", "frac"), step_gcups=_pick(c1, "step_roofline", "valu", "achieved_gcups"),
            traffic=_pick(c1, "step_roofline", "traffic"), alg_bytes_per_launch=_pick(c1, "step_roofline", "algorithmic_bytes_per_launch"),
            traffic_scope="phases A+B (step_roofline), per traced launch" if _pick(c1, "step_roofline", "traffic") else None,
            cpu_reads_per_s=_pick(c1, "cpu_baseline", "value"))
    c2 = also.get("configs2", {})
    if c2:
        leg("configs2", failed=c2.get("failed"), reads_per_s=c2.get("reads_per_s"), ms_per_step=c2.get("ms_per_step"),
            pairs_per_read=c2.get("pairs_per_read"), binned_to_planted=c2.get("reads_binned_to_their_planted_barcode"),
            pruned_reads_per_s=_pick(c2, "exact_pruning", "reads_per_s"), pruned_ms_per_step=_pick(c2, "exact_pruning", "ms_per_step"),
            pruned_same=_pick(c2, "exact_pruning", "same_trims_and_calls_as_tracing_every_pair"),
            pruned_traced_fraction=_pick(c2, "exact_pruning", "pairs_traced_fraction"),
            score_pass_tcups=_pick(c2, "exact_pruning", "score_pass", "tcups"),
            parity_checked=_pick(c2, "parity", "checked"), mismatches=_pick(c2, "parity", "mismatches"),
            crosscheck_pairs=_pick(c2, "parity", "device_crosscheck", "device_crosscheck_pairs"),
            crosscheck_differing=_pick(c2, "parity", "device_crosscheck", "records_differing"),
            roofline_frac=_pick(c2, "roofline", "frac"), valu_frac=_pick(c2, "roofline", "valu", "frac"),
            avg_launch_ms=_pick(c2, "roofline", "avg_launch_ms"), traffic=_pick(c2, "roofline", "traffic"),
            alg_bytes_per_launch=_pick(c2, "roofline", "algorithmic_bytes_per_launch"),
            cpu_reads_per_s=_pick(c2, "cpu_baseline", "value"))
    c4 = also.get("configs4_per_gpu", {})
    if c4:
        leg("configs4_per_gpu", failed=c4.get("failed"), n_gpus=c4.get("n_gpus"), reads_per_gpu=c4.get("reads_per_gpu"),
            reads_per_s=c4.get("reads_per_s"), ms_per_step=c4.get("ms_per_step"), middle_adapters=c4.get("middle_adapters"),
            binned_to_planted=c4.get("reads_binned_to_their_planted_barcode"),
            fast_reads_per_s=_pick(c4, "exact_prefilter", "reads_per_s"), fast_ms_per_step=_pick(c4, "exact_prefilter", "ms_per_step"),
            fast_same=_pick(c4, "exact_prefilter", "same_trims_calls_and_middle_hits"),
            kernels_compiled=_pick(c4, "specialised_kernels", "compiled_in_this_process"),
            kernels_loaded=_pick(c4, "specialised_kernels", "loaded_from_the_kernel_cache"),
            parity_checked=_pick(c4, "parity", "checked"), mismatches=_pick(c4, "parity", "mismatches"),
            roofline_frac=_pick(c4, "roofline", "frac"), valu_frac=_pick(c4, "roofline", "valu", "frac"),
            avg_launch_ms=_pick(c4, "roofline", "avg_launch_ms"), traffic=_pick(c4, "roofline", "traffic"),
            seed_scan_hbm_frac=_pick(c4, "exact_prefilter", "roofline", "frac"),
            cpu_reads_per_s=_pick(c4, "cpu_baseline", "value"))
    pf = cfg.get("exact_prefilter", {})
    if pf:
        leg("exact_prefilter", reads_per_s=pf.get("reads_per_s"), ms_per_step=pf.get("ms_per_step"),
            same=pf.get("same_trims_and_middle_hits"), packed_ms_per_step=_pick(pf, "reads_resident_at_2_bits_per_base", "ms_per_step"),
            packed_same=_pick(pf, "reads_resident_at_2_bits_per_base", "same_trims_and_middle_hits"),
            packed_seed_scan_ms=_pick(pf, "reads_resident_at_2_bits_per_base", "seed_scan_ms"),
            packed_hbm_bytes=_pick(pf, "reads_resident_at_2_bits_per_base", "hbm_bytes_resident"),
            packed_failed=_pick(pf, "reads_resident_at_2_bits_per_base", "failed"),
            seed_scan_gb_per_s=_pick(pf, "roofline", "achieved"),
            seed_scan_hbm_frac=_pick(pf, "roofline", "frac"), seed_scan_ms=_pick(pf, "roofline", "avg_launch_ms"),
            seed_scan_traffic=_pick(pf, "roofline", "traffic"), seed_scan_alg_bytes=_pick(pf, "roofline", "algorithmic_bytes_per_launch"))
    pr = cfg.get("optional_exact_pruning", {})
    if pr:
        leg("proven_middle_scan", reads_per_s=pr.get("reads_per_s"), ms_per_step=pr.get("ms_per_step"), same=pr.get("same_middle_hits"))
    rg = also.get("ragged_lengths", {})
    if rg:
        leg("ragged_lengths", failed=rg.get("failed"), reads_per_s=rg.get("reads_per_s"), read_bp_per_s=rg.get("read_bp_per_s"),
            bp_vs_uniform=rg.get("bp_per_s_vs_uniform_lengths"), parity_checked=_pick(rg, "parity", "checked"),
            mismatches=_pick(rg, "parity", "mismatches"))
    ul = also.get("ultralong", {})
    if ul:
        leg("ultralong", failed=ul.get("failed"), reads=ul.get("reads"), max_len=ul.get("max_len"), reads_over_65535=ul.get("reads_over_65535"),
            reads_per_s=ul.get("reads_per_s"), read_bp_per_s=ul.get("read_bp_per_s"), ms_per_step=ul.get("ms_per_step"),
            prefiltered_read_bp_per_s=ul.get("prefiltered_read_bp_per_s"), same=ul.get("prefiltered_same"),
            hits_beyond_65535=ul.get("middle_hits_beyond_column_65535"), parity_checked=_pick(ul, "parity", "checked"),
            mismatches=_pick(ul, "parity", "mismatches"), longest_checked_bp=_pick(ul, "parity", "longest_checked_bp"))
    hb = also.get("from_host_memory", {})
    if hb:
        leg("from_host_memory", failed=hb.get("failed"), reads_per_s=hb.get("reads_per_s"), ms_per_step=hb.get("ms_per_step"),
            h2d_gb_per_s=hb.get("h2d_gb_per_s_alone"), h2d_ms=hb.get("h2d_ms_alone"), packed_2bit=hb.get("packed"),
            bytes_per_step=hb.get("bytes_uploaded_per_step"), pack_once_s=hb.get("pack_once_s"),
            unpacked_reads_per_s=_pick(hb, "bytes_per_base_1", "reads_per_s"), unpacked_h2d_ms=_pick(hb, "bytes_per_base_1", "h2d_ms_alone"),
            same_hits_both_forms=hb.get("same_hits_both_forms"), prefiltered_reads_per_s=_pick(hb, "exact_prefilter", "reads_per_s"),
            prefiltered_same_hits=_pick(hb, "exact_prefilter", "same_hits"),
            kept_packed_reads_per_s=_pick(hb, "exact_prefilter", "kept_packed", "reads_per_s"),
            kept_packed_same_hits=_pick(hb, "exact_prefilter", "kept_packed", "same_hits"),
            kept_packed_failed=_pick(hb, "exact_prefilter", "kept_packed", "failed"))
    ee = also.get("end_to_end", {})
    if ee:
        leg("end_to_end", failed=ee.get("failed"), reads_per_s=ee.get("reads_per_s"), wall_s=ee.get("wall_s"),
            input_gb_per_s=ee.get("input_gb_per_s"), streamed_equals_whole=ee.get("streamed_output_identical_to_whole_file_output"),
            whole_file_reads_per_s=_pick(ee, "whole_file_path", "reads_per_s"))
    ft = also.get("configs4_fixed_total", {})
    if ft:
        leg("configs4_fixed_total", scaling=ft.get("scaling"), n_gpus=ft.get("n_gpus"), reads_total=ft.get("reads_total"),
            reduced_to=ft.get("reads_total_reduced_to_fit_host_memory"), reads_per_s=ft.get("reads_per_s"), wall_s=ft.get("wall_s"),
            ms_by_rank=[_r(x) for x in ft.get("ms_by_rank") or []], host_threads_per_rank=ft.get("host_threads_per_rank"),
            world_size_seen=ft.get("world_size_seen"), backend=ft.get("backend"), check_reads_by_rank=ft.get("check_reads_by_rank"),
            fast_same=ft.get("fast_same"))
    sf = also.get("sharded_file", {})
    if sf:
        leg("sharded_file", n_gpus=sf.get("n_gpus"), reads_per_s=sf.get("reads_per_s"), wall_s=sf.get("wall_s"),
            reads_by_rank=sf.get("reads_by_rank"), md5_equal=sf.get("md5_equal"))
    eg = also.get("end_to_end_gz", {})
    if eg:
        leg("end_to_end_gz", failed=eg.get("failed"), reads_per_s=eg.get("reads_per_s"), wall_s=eg.get("wall_s"),
            single_member_reads_per_s=eg.get("single_member_reads_per_s"), cat_members_reads_per_s=eg.get("concatenated_members_reads_per_s"),
            md5_equal=eg.get("md5_equal"),
            deflate_gb_per_s=eg.get("deflate_gb_per_s"), deflate_cores=eg.get("deflate_cores"), deflate_ratio=eg.get("deflate_ratio"),
            inflate_gb_per_s=eg.get("inflate_sized_members_gb_per_s"), ref_cli_reads_per_s=_pick(eg, "reference_cli", "reads_per_s"),
            ref_cli_compressor=_pick(eg, "reference_cli", "compressor"), ref_cli_md5_equal=_pick(eg, "reference_cli", "md5_equal"),
            speedup_vs_reference_cli=eg.get("speedup_vs_reference_cli"))
    b1 = cpu.get("b1_cli") or {}
    dr = full.get("dropin") or {}
    par = full.get("parity") or {}
    flat = {}
    short = {"configs1": "c1", "configs2": "c2", "configs4_per_gpu": "c4", "exact_prefilter": "pf", "ragged_lengths": "ragged",
             "from_host_memory": "h2d", "end_to_end": "e2e", "proven_middle_scan": "proven", "ultralong": "ul", "end_to_end_gz": "e2egz", "configs4_fixed_total": "c4total", "sharded_file": "sharded"}
    for name, d in legs.items():
        for k in ("reads_per_s", "parity_checked", "mismatches", "roofline_frac", "valu_frac", "valu_frac_nominal", "cpu_reads_per_s",
                  "pruned_reads_per_s", "fast_reads_per_s", "seed_scan_hbm_frac", "same", "pruned_same", "fast_same", "md5_equal",
                  "deflate_gb_per_s"):
            if k in d:
                flat["%s_%s" % (short[name], k)] = d[k]
    out = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                        "scaling", "vs_baseline", "dtype", "data", "read_bp_per_s", "value_incl_h2d")}
    out["config"] = {"workload": cfg.get("workload"), "reads_per_gpu": cfg.get("reads_per_gpu"), "read_len": cfg.get("read_len"),
                     "parallelism": cfg.get("parallelism"), "world_size": cfg.get("world_size"), "backend": cfg.get("backend"),
                     "library_sha1": cfg.get("library_sha1"), "middle_hits_per_step": cfg.get("middle_hits_per_step"),
                     "mask_rounds": cfg.get("mask_rounds"), "matching_sets": ",".join(cfg.get("matching_sets") or []),
                     "speedup_vs_cpu_baseline": _r(cfg.get("speedup_vs_cpu_baseline")),
                     "speedup_vs_reference_cli": _r(full["value"] / b1["best_reads_per_s"]) if b1.get("best_reads_per_s") else None,
                     "region_ms_min": _r(_pick(full, "repeats", "min")), "region_ms_median": _r(_pick(full, "repeats", "median")),
                     "region_ms_max": _r(_pick(full, "repeats", "max")),
                     "ms_per_step_by_rank": [_r(x) for x in cfg.get("ms_per_step_by_rank") or []],
                     "check_reads": cfg.get("check_reads"), "check_reads_by_rank": cfg.get("check_reads_by_rank"),
                     "rccl_world_size_seen": cfg.get("rccl_world_size_seen"), "device_by_rank": cfg.get("device_by_rank"),
                     "gpus_visible": cfg.get("gpus_visible"), "self_launched": cfg.get("self_launched")}
    out["config"].update({"kernel_ms_" + k: _r(v) for k, v in (cfg.get("kernel_ms_per_step") or {}).items() if v})
    out["config"].update(flat)
    if dr:
        out["config"].update({"dropin_reads_per_s": _r(dr.get("reads_per_s")), "dropin_md5_equal": dr.get("md5_equal"),
                              "dropin_misses": dr.get("misses")})
    out["roofline"] = None
    if roof:
        out["roofline"] = {"bound": roof.get("bound"), "kernel": (roof.get("kernel") or "").split(" ")[0],
                           "achieved": _r(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"),
                           "frac": _r(roof.get("frac")), "traffic": _r(roof.get("traffic")),
                           "traffic_measured_in_this_run": bool(roof.get("traffic_measured_in_this_run")),
                           "traffic_from_profile": _r(roof.get("traffic_from_profile")),
                           "traffic_source": (roof.get("traffic_source") or "").split(" ")[0] or None,
                           "launches": roof.get("launches"), "avg_launch_ms": _r(roof.get("avg_launch_ms")),
                           "algorithmic_bytes_per_launch": _r(roof.get("algorithmic_bytes_per_launch")),
                           "valu_gcups": _r(_pick(roof, "valu", "achieved_gcups")), "valu_peak_gcups": _r(_pick(roof, "valu", "peak_gcups")),
                           "valu_frac": _r(_pick(roof, "valu", "frac")), "ops_per_2_cells": _pick(roof, "valu", "ops_per_2_cells"),
                           "valu_frac_nominal": _r((_pick(roof, "valu", "frac") or 0.0) / 2.0),
                           "valu_peak_nominal_gcups": _r((_pick(roof, "valu", "peak_gcups") or 0.0) * 2.0),
                           "valu_nominal_is": "the guide's 2-cycle wave64 issue: 1.2 G wave-instr/s/SIMD at 2.4 GHz"}
        if roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
            out["roofline"]["traffic_ratio"] = _r(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 4)
    if cpu:
        out["cpu_baseline"] = {"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                               "sample": cpu.get("sample")}
        if b1:
            c = out["cpu_baseline"]
            c["b1_cli_kind"] = b1.get("kind", "reference CLI")
            c["b1_cli_failed"] = b1.get("failed")
            c["b1_cli_reads"] = b1.get("reads")
            for k, v in b1.items():
                if k.startswith("threads_") and isinstance(v, dict):
                    c["b1_cli_%s_reads_per_s" % k] = _r(v.get("reads_per_s"))
            c["b1_cli"] = {k: ({kk: (_r(vv) if not isinstance(vv, dict) else {a: _r(b) for a, b in vv.items()}) for kk, vv in v.items()}
                               if isinstance(v, dict) else _r(v)) for k, v in b1.items()}
            out["cpu_baseline"] = {k: v for k, v in c.items() if v is not None}
    out["parity"] = {"checked": par.get("checked"), "mismatches": par.get("mismatches"),
                     "phase_a_reads": _pick(par, "phase_a_rederived_on_cpu", "reads"),
                     "phase_a_entries_differing": _pick(par, "phase_a_rederived_on_cpu", "entries_differing"),
                     "phase_a_same_matching_sets": _pick(par, "phase_a_rederived_on_cpu", "same_matching_sets"),
                     "crosscheck_pairs": _pick(par, "device_crosscheck", "device_crosscheck_pairs"),
                     "crosscheck_differing": _pick(par, "device_crosscheck", "records_differing"),
                     "runner_vs_reference_cli_reads": _pick(par, "runner_vs_reference_cli", "reads"),
                     "runner_vs_reference_cli_md5_equal": _pick(par, "runner_vs_reference_cli", "md5_equal")}
    if dr:
        out["dropin"] = {k: ({kk: (_r(vv) if not isinstance(vv, dict) else {a: _r(b) for a, b in vv.items()}) for kk, vv in v.items()}
                             if isinstance(v, dict) else _r(v)) for k, v in dr.items()}
    out["legs"] = legs
    line = json.dumps(out, separators=(",", ":"))
    flat_nominal = [("config", k) for k in list(out["config"]) if k.endswith("_valu_frac_nominal")]
    for victim in [("roofline", "valu_nominal_is"), ("legs", "configs1", "traffic_scope")] + flat_nominal + \
            [("dropin", "threads_1", "phase_s"), ("cpu_baseline", "b1_cli", "threads_1", "phase_s"), ("config", "ms_per_step_by_rank"),
             ("dropin", "threads_16", "phase_s"), ("cpu_baseline", "b1_cli", "threads_16", "phase_s")]:
        if len(line) <= 7600:
            break
        d = out
        for k in victim[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict):
            d.pop(victim[-1], None)
        line = json.dumps(out, separators=(",", ":"))
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """Re-execute this very command line as n ranks of one node (torch.distributed.run: one process per GPU over RCCL;
    PC_DIST_BACKEND=gloo lets more ranks than GPUs share one for functional checks).  -> the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env["PC_BENCH_SELF_LAUNCHED"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (headline, configs[3])")
    ap.add_argument("--reads1", type=int, default=100_000, help="reads of the configs[1] leg")
    ap.add_argument("--reads2", type=int, default=1_000_000, help="reads of the configs[2] leg")
    ap.add_argument("--reads4", type=int, default=1_250_000, help="reads per GPU of the configs[4]-shape leg (10 M over 8 GPUs)")
    ap.add_argument("--reads-e2e", type=int, default=400_000, help="reads of the end-to-end (file -> file) leg")
    ap.add_argument("--reads4-total", type=int, default=10_000_000, help="N > 1 only: the FIXED total of BASELINE configs[4] split over the ranks (strong scaling, from host memory)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps (the first one is `value`)")
    ap.add_argument("--read-len", type=int, default=8000)
    ap.add_argument("--chimera", type=float, default=0.01)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the CPU baseline / parity legs")
    ap.add_argument("--no-extra", action="store_true", help="headline only (skip the configs[1] / configs[2] legs)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc passes that measure roofline.traffic in this run")
    ap.add_argument("--cli-reads", type=int, default=2000, help="reads of the reference-CLI baseline (B1); 0 disables it and the drop-in leg")
    ap.add_argument("--dropin-reads", type=int, default=20000, help="reads of the drop-in leg (unchanged reference Python over the HIP library)")
    ap.add_argument("--dropin-threads", type=int, default=16, help="--threads of the drop-in leg's second run (the first uses 1)")
    ap.add_argument("--full-json", default=os.environ.get("PC_BENCH_FULL", ""), help="also write the unabridged record here "
                    "(default: gpurun_out/bench_full.json under the repo)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it starts its own N ranks: the same command re-executed
    # under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1); rank 0 prints the one line.  Under
    # torchrun (WORLD_SIZE set) nothing is re-launched.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU over RCCL (backend "nccl" IS RCCL on ROCm).  PC_DIST_BACKEND=gloo with more
    # ranks than GPUs is only for functional checks of the N>1 path on a single-GPU box.
    backend = os.environ.get("PC_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: measuring %d" % (args.gpus, world, world), file=sys.stderr)
    if os.environ.get("PC_BENCH_LAUNCH_PROBE", "0") not in ("", "0"):
        # launch plumbing only (tests/test_bench_launch_cpu.py, no GPU needed): every rank reports in, rank 0 prints who came
        seen = [None] * world
        if world > 1:
            dist.all_gather_object(seen, (rank, local_rank, os.getpid()))
        else:
            seen = [(rank, local_rank, os.getpid())]
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1,
                              "backend": dist.get_backend() if world > 1 else "none", "ranks": [list(x) for x in seen],
                              "self_launched": os.environ.get("PC_BENCH_SELF_LAUNCHED", "0") == "1", "steps": args.steps}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads

    params = ScanParams()
    pl = Pipeline(load_panel_sets(), params, device=dev)
    # seed 3 = BASELINE config 4; every rank draws its own shard (rank-dependent seed)
    reads = make_reads(args.reads, args.read_len, seed=3 + 1000 * rank, start_frac=0.9, end_frac=0.5,
                       chimera_frac=args.chimera, device=dev)
    # each rank checks its share of the first 10 000 (porechop.py:86 --check_reads): the shares sum to exactly that
    n_check = params.check_reads // world + (1 if rank < params.check_reads % world else 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def region(steps, **kw):
        """One timed region of the contract: barrier + synchronize, `steps` steps, barrier + synchronize; MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = one_step(pl, reads, n_check, world, **kw)
        pl.aligner.sync()
        barrier()
        tall = torch.zeros(world, dtype=torch.float64, device=dev)
        tall[rank] = time.perf_counter() - t0
        if world > 1:
            dist.all_reduce(tall, op=dist.ReduceOp.SUM)
        rank_s = [float(x) for x in tall.cpu()]
        return out, max(rank_s), rank_s

    shares = torch.zeros(world, dtype=torch.int64, device=dev)
    shares[rank] = min(n_check, args.reads)
    if world > 1:
        dist.all_reduce(shares, op=dist.ReduceOp.SUM)
    check_shares = [int(x) for x in shares.cpu()]
    devs = torch.zeros(world, dtype=torch.int64, device=dev)
    devs[rank] = dev_index
    if world > 1:
        dist.all_reduce(devs, op=dist.ReduceOp.SUM)
    device_by_rank = [int(x) for x in devs.cpu()]
    matching = None
    for _ in range(args.warmup):
        matching, st, et, hits = one_step(pl, reads, n_check, world)
        pl.aligner.sync()
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    pl.stats = {k: 0 for k in pl.stats}
    # ---- THE measurement: exactly --steps steps ----------------------------------------------------------
    (matching, st, et, hits), dt, rank_s = region(args.steps)
    timing = pl.aligner.get_timing()
    pl.aligner.set_timing(False)

    # ---- AFTER it: the same region twice more (run-to-run spread; `value` stays the first region's) --------
    region_ms = [dt / args.steps * 1e3]
    for _ in range(max(0, args.repeats - 1)):
        _, dtr, _ = region(args.steps)
        region_ms.append(dtr / args.steps * 1e3)

    # ---- the same steps with the two optional exact prunings (score bound for the identity thresholds:
    # fewer tracebacks, identical sets / trims / hits).  Reported as an extra field; never `value`.
    psteps = max(1, min(args.steps, 5))
    one_step(pl, reads, n_check, world, proofs=True)
    (_, _, _, hits_p), dt_proofs, _ = region(psteps, proofs=True)
    same_hits = bool(hits_p.read.numel() == hits.read.numel() and torch.equal(hits_p.read, hits.read) and
                     torch.equal(hits_p.start, hits.start) and torch.equal(hits_p.end, hits.end))

    # ---- the same steps with the exact bit-parallel prefilter in front of the middle scan (the
    # alternative the reference's README.md:355-357 names): pairs farther than max_edits(m, threshold) from every
    # substring are proven non-hits and never reach the DP.  Reported BESIDE the headline, never as `value`.
    one_step(pl, reads, n_check, world, prefilter=True)
    pl.aligner.set_timing(True)
    pl.aligner.get_timing()
    fsteps = max(1, min(args.steps, 10))
    (_, st_f, et_f, hits_f), dt_pf, _ = region(fsteps, prefilter=True)
    timing_pf = pl.aligner.get_timing()
    pl.aligner.set_timing(False)
    pf_regions = [dt_pf / fsteps * 1e3]
    for _ in range(max(0, args.repeats - 1)):          # this variant has a host round trip per step: report its spread too
        _, dtr, _ = region(fsteps, prefilter=True)
        pf_regions.append(dtr / fsteps * 1e3)
    dt_pf = sorted(pf_regions)[len(pf_regions) // 2] * fsteps / 1e3          # the MEDIAN region is the one reported
    same_hits_f = bool(hits_f.read.numel() == hits.read.numel() and torch.equal(hits_f.read, hits.read) and
                       torch.equal(hits_f.adapter, hits.adapter) and torch.equal(hits_f.start, hits.start) and
                       torch.equal(hits_f.end, hits.end) and torch.equal(st_f, st) and torch.equal(et_f, et) and
                       (hits_f.rounds, hits_f.alignments) == (hits.rounds, hits.alignments))

    # ---- the same step with the reads RESIDENT AT 2 BITS PER BASE (north_star: "2-bit-packed read windows"): the plane
    # io.pack_reads makes stays in HBM (a quarter of the bytes), the prefilter's seed scan reads it directly
    # (pc_prefilter_packed), and only the two 150-base end windows of every read and the few reads that survive the
    # prefilter are ever turned into bytes (pc_unpack_windows).  Beside the headline, never `value`.
    pk_out = None
    if world == 1 and not args.no_extra:
        try:
            from porechop_amd.io import pack_reads
            from porechop_amd.pipeline import DeviceReads
            nb_ = int(reads.off[-1].item()) + int(reads.length[-1].item())
            h_ = reads.arena[:nb_].cpu().numpy()
            pk_, exc_ = pack_reads(h_, nb_)
            del h_
            t0p = time.perf_counter()
            reads_pk = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk_).to(dev), nb_,
                                               torch.from_numpy(exc_).to(dev) if exc_.size else None, reads.off, reads.length,
                                               end_size=params.end_size)
            pl.aligner.sync()
            torch.cuda.synchronize()
            build_ms = (time.perf_counter() - t0p) * 1e3
            for _ in range(2):                       # (the first step over a new resident form sizes the library's scratch buffers)
                one_step(pl, reads_pk, n_check, world, prefilter=True)
            pl.aligner.set_timing(True)
            pl.aligner.get_timing()
            stats0 = dict(pl.stats)
            barrier()
            t0p = time.perf_counter()
            for _ in range(fsteps):
                _, st_k, et_k, hits_k = one_step(pl, reads_pk, n_check, world, prefilter=True)
            pl.aligner.sync()
            barrier()
            dt_pk = time.perf_counter() - t0p
            timing_pk = pl.aligner.get_timing()
            pl.aligner.set_timing(False)
            same_k = bool(hits_k.read.numel() == hits.read.numel() and torch.equal(hits_k.read, hits.read) and
                          torch.equal(hits_k.adapter, hits.adapter) and torch.equal(hits_k.start, hits.start) and
                          torch.equal(hits_k.end, hits.end) and torch.equal(st_k, st) and torch.equal(et_k, et))
            pk_out = {"reads_per_s": args.reads * fsteps / dt_pk, "ms_per_step": dt_pk / fsteps * 1e3, "same_trims_and_middle_hits": same_k,
                      "hbm_bytes_resident": int(reads_pk.plane.numel()) + int(reads_pk.ends[0].numel()),
                      "hbm_bytes_resident_as_bytes": int(reads.arena.numel()),
                      "bytes_unpacked_per_step": (pl.stats.get("bases_unpacked_after_prefilter", 0) - stats0.get("bases_unpacked_after_prefilter", 0)) // fsteps,
                      "packed_route_refused": pl.stats.get("packed_route_refused", 0) - stats0.get("packed_route_refused", 0),
                      "upload_and_end_windows_ms": build_ms,
                      "kernel_ms_per_step": {k: v[0] / fsteps for k, v in timing_pk.items()},
                      "seed_scan_ms": timing_pk["seed_scan"][0] / max(1, timing_pk["seed_scan"][1]) if timing_pk["seed_scan"][1] else None,
                      "seed_scan_plane_gb_per_s": (nb_ / 4 / 1e9) / (timing_pk["seed_scan"][0] / 1e3 / fsteps) if timing_pk["seed_scan"][0] else None}
            del reads_pk
        except Exception as e:      # an extra measurement must never break the bench line
            pk_out = {"failed": repr(e)}

    out = None
    if rank == 0:
        total_reads = args.reads * world
        reads_per_s = total_reads * args.steps / dt
        # ---- roofline of the dominant kernel: the score-only whole-read scan ---------------
        # algorithmic bytes: per read |H| input bytes ONCE for all A middle
        # adapters + 28 B of result per (read, adapter); a launch scanning one of A adapters is
        # credited 1/A of the read bytes.  cells = sum |H| x |V|.
        # Dominant kernel = the specialised scan (timing kind 'score_spec'); the generic
        # ahead-of-time one ('score') only if specialisation is off.  One timed region per launch,
        # so avg_launch_ms is directly rocprofv3's AverageNs for that kernel name.
        jit = timing["score_spec"][1] > 0
        ms, launches, pairs = timing["score_spec"] if jit else timing["score"]
        A = max(1, len(pl.middle_adapter_list(matching)))
        mean_trim_len = float((reads.length.to(torch.float64) - st.to(torch.float64) - et.to(torch.float64)).mean().item())
        roof = None
        if launches > 0:
            per_launch_s = ms / 1e3 / launches
            pairs_per_launch = pairs / launches
            alg_bytes = pairs_per_launch * (mean_trim_len / A + 28.0)
            achieved = alg_bytes / per_launch_s / 1e9
            mean_m = float(np.mean([len(a[1]) for a in pl.middle_adapter_list(matching)]))
            cells = pairs_per_launch * mean_trim_len * mean_m
            # the specialised score kernel spends 5 packed-fp16 ops per 2 cells (6 in its int16
            # variant; generic kernel: 9)
            ops_per_pair = (6 if os.environ.get("PC_JIT_INT16", "0") not in ("", "0") else 5) if jit else 9
            valu_peak_gcups = VALU_WAVE_INSTR_PER_S * 64 * 2 / ops_per_pair / 1e9
            roof = {"bound": "valu",
                    "kernel": ("pc_spec_score (specialised score-only whole-read scan)" if jit
                               else "scan_kernel<R,PAD,false> (generic score-only whole-read scan)"),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "launches": int(launches), "avg_launch_ms": per_launch_s * 1e3,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "gcups": cells / per_launch_s / 1e9,
                    "valu": {"achieved_gcups": cells / per_launch_s / 1e9, "peak_gcups": valu_peak_gcups,
                             "frac": cells / per_launch_s / 1e9 / valu_peak_gcups, "ops_per_2_cells": ops_per_pair,
                             "peak_source": "one wave64 VALU instruction per 4 cycles per SIMD x 1024 SIMDs x 2.4 GHz "
                                            "(tools/ubench_valu.hip, profiles/r02_ubench_valu.txt)"},
                    "note": "achieved/peak/frac are the HBM roofline on ALGORITHMIC bytes as the contract asks; the kernel is "
                            "integer max-plus DP with >= 50 cells per algorithmic byte, i.e. VALU-bound by construction "
                            "(bound: valu) and cannot approach the HBM roofline; the launch average includes the small "
                            "mask-and-realign launches of the same kernel (DESIGN.md section 4)"}
            # HBM-side traffic of that kernel comes from the separately collected rocprofv3 --pmc passes
            # of this same command (tools/profile_round.sh -> profiles/<round>_summary.json): FETCH_SIZE +
            # WRITE_SIZE of one launch, KB as reported -- accepted only when that summary was taken with the
            # library now loaded (profile_traffic).  (The guide's x2 correction is for wide 16-B/lane
            # streams; here lanes gather 4 B each and the kernel provably consumes 7.95 GB per launch
            # against a reported FETCH_SIZE of 6.4 GB, so no doubling is applied.)
            if world == 1:
                roof["traffic"] = profile_traffic("headline", "pc_spec_score" if jit else "scan_kernel<score>", launches / args.steps)
                if roof["traffic"] is not None:
                    roof["traffic_source"] = _PROFILE.get("_path", "") + " (mean over this kernel's launches; FETCH_SIZE + WRITE_SIZE; same library sha1)"
                # ... and measured IN THIS RUN where the box has rocprofv3: two --pmc passes (one counter each, nothing else enabled)
                # over a child process that runs two headline steps and nothing else (tools/run_leg.py) -- the guide's recipe,
                # spawned from here so that the driver's own line carries counters of the library it ran
                if not args.no_extra and not args.no_live_traffic:
                    live = live_traffic("pc_spec_score" if jit else "scan_kernel")
                    if live:
                        roof["traffic_from_profile"] = roof["traffic"]
                        roof["traffic"] = live["bytes_per_launch"]
                        roof["traffic_measured_in_this_run"] = True
                        roof["traffic_live"] = live
                        roof["traffic_source"] = "this-run (rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE, over `python tools/run_leg.py headline 2`, spawned by bench.py)"
        kern_ms = {k: v[0] / args.steps for k, v in timing.items()}
        srt = sorted(region_ms)
        pf = {"reads_per_s": total_reads * fsteps / dt_pf, "ms_per_step": dt_pf / fsteps * 1e3, "steps": fsteps,
              "same_trims_and_middle_hits": same_hits_f, "speedup_vs_headline": (dt / args.steps) / (dt_pf / fsteps),
              "ms_per_step_by_region": pf_regions,
              "kernel_ms_per_step": {k: v[0] / fsteps for k, v in timing_pf.items()},
              "pairs_reaching_the_dp_per_step": pl.stats.get("pairs_middle_scanned_after_prefilter", 0) // max(1, fsteps * len(pf_regions) + 1),
              "pairs_prefiltered_per_step": pl.stats.get("pairs_middle_prefiltered", 0) // max(1, fsteps * len(pf_regions) + 1),
              "note": "not the headline: every (read, adapter) pair of the middle scan first goes through the exact prefilter -- a hit "
                      "needs the adapter within max_edits(m, --middle_threshold) unit-cost edits of a substring of the read; that is "
                      "decided exactly by one HBM-bound pass that finds the exact q-gram seeds such an occurrence must contain, Myers' "
                      "bit-vector edit distance on the finds -- and only the surviving pairs run the DP; everything else is proven "
                      "not to be a hit (csrc/pc_prefilter.hip, tests/test_prefilter_bound.py, tests/test_gpu_prefilter.py)"}
        pf["roofline"] = prefilter_roofline(timing_pf, mean_trim_len, A)
        if pk_out is not None:
            pf["reads_resident_at_2_bits_per_base"] = pk_out
        out = {
            "metric": "reads/sec (and read-bp/sec) end+middle adapter scan, 8 kb reads",
            "value": reads_per_s, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "integers held exactly in packed fp16 lanes (pc_spec_score and trace16_kernel: every value within +-2040, gated on "
                     "the host and asserted on the device); packed int16 / plain int32 kernels for schemes outside that range", "data": "synthetic",
            "read_bp_per_s": reads_per_s * args.read_len,
            "repeats": {"ms_per_step": region_ms, "median": srt[len(srt) // 2], "min": srt[0], "max": srt[-1],
                        "note": "%d timed regions of %d steps each, back to back; `value` is the FIRST (the contract's) region"
                                % (len(region_ms), args.steps)},
            "config": {"workload": "BASELINE configs[3] shape: %d synthetic %d-bp reads per GPU, %.0f%% chimeras, phases A (119-set "
                                   "panel, %d check reads) + B + C (middle scan on); generator porechop_amd/synth.py: bodies from a "
                                   "torch generator, adapter copies drawn from a pool of 4096 pre-mutated instances"
                                   % (args.reads, args.read_len, args.chimera * 100, params.check_reads),
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "parallelism": "reads sharded x%d" % world,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "backend": (dist.get_backend() if world > 1 else "none (single process)"),
                       "rccl_world_size_seen": (dist.get_world_size() if world > 1 and dist.get_backend() == "nccl" else None),
                       "device_by_rank": device_by_rank, "gpus_visible": torch.cuda.device_count(),
                       "self_launched": os.environ.get("PC_BENCH_SELF_LAUNCHED", "0") == "1",
                       "ms_per_step_by_rank": [s_ / args.steps * 1e3 for s_ in rank_s],
                       "check_reads": params.check_reads, "check_reads_by_rank": check_shares,
                       "matching_sets": [pl.sets[i].name for i in matching],
                       "middle_hits_per_step": int(hits.read.numel()), "mask_rounds": hits.rounds,
                       "kernel_ms_per_step": kern_ms,
                       "library_sha1": library_fingerprint(),
                       "optional_exact_pruning": {"reads_per_s": total_reads * psteps / dt_proofs,
                                                  "ms_per_step": dt_proofs / psteps * 1e3, "same_middle_hits": same_hits,
                                                  "note": "not the headline: phase A and the middle scan trace back only "
                                                          "pairs whose score can still reach the identity threshold "
                                                          "(DESIGN.md section 7)"},
                       "exact_prefilter": pf},
            "roofline": roof,
        }
        if args.cpu_seconds > 0 and world == 1:
            try:
                out["cpu_baseline"], out["parity"] = cpu_baseline(reads, pl, matching, st, et, hits, args.cpu_seconds, host_cores())
                out["config"]["speedup_vs_cpu_baseline"] = reads_per_s / out["cpu_baseline"]["value"]
            except Exception as e:   # the baseline leg must never break the bench line
                out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
            try:
                out.setdefault("parity", {})["phase_a_rederived_on_cpu"] = cpu_phase_a_check(reads, pl, host_cores())
            except Exception as e:
                out.setdefault("parity", {})["phase_a_rederived_on_cpu"] = {"failed": repr(e)}
        if args.cpu_seconds > 0 and world == 1 and args.cli_reads > 0:
            from tests.ref_cli import staged
            if staged():
                import shutil
                import tempfile
                work = tempfile.mkdtemp(prefix="pc_bench_cli_")
                try:
                    cli = fq = None
                    try:
                        note("leg reference_cli")
                        cli, fq = leg_reference_cli(reads, args, host_cores(), work)
                        out["cpu_baseline"]["b1_cli"] = cli
                    except Exception as e:
                        out["cpu_baseline"]["b1_cli"] = {"failed": repr(e)}
                    try:
                        # the batch runner (no reference Python at all) on the same file: the reference CLI's output, byte for byte
                        from porechop_amd import runner
                        o_run = os.path.join(work, "runner_out.fastq")
                        t0 = time.perf_counter()
                        runner.run(fq, output=o_run, device=dev)
                        out.setdefault("parity", {})["runner_vs_reference_cli"] = {
                            "reads": cli["reads"], "md5_equal": bool(file_md5(o_run) == cli["output_md5"]), "runner_s": time.perf_counter() - t0}
                    except Exception as e:
                        out.setdefault("parity", {})["runner_vs_reference_cli"] = {"failed": repr(e)}
                    try:
                        note("leg dropin")
                        if fq is None:
                            fq = os.path.join(work, "cli_reads.fastq")
                            write_fastq(reads, min(args.cli_reads, reads.n), fq)
                        out["dropin"] = leg_dropin(reads, args, work, cli, fq)
                    except Exception as e:
                        out["dropin"] = {"failed": repr(e)}
                finally:
                    shutil.rmtree(work, ignore_errors=True)
            else:
                out["cpu_baseline"]["b1_cli"] = {"failed": "no staged reference under oracle/_ref/porechop_ref (make -C oracle ref)"}
        if world == 1 and not args.no_extra:
            try:
                out.setdefault("parity", {})["device_crosscheck"] = device_crosscheck(pl, reads, matching, st, et, dev)
            except Exception as e:
                out.setdefault("parity", {})["device_crosscheck"] = {"failed": repr(e)}
    pl.close()
    del reads, pl
    torch.cuda.empty_cache()
    also = {}
    if not args.no_extra:
        # the per-GPU shape of BASELINE configs[4], on EVERY rank (its own barriers / MAX over ranks inside)
        note("leg configs4_per_gpu")
        try:
            r4 = leg_configs4(dev, args, host_cores(), world, rank, barrier)
            if rank == 0:
                also["configs4_per_gpu"] = r4
        except Exception as e:
            if world > 1:
                raise                                    # a rank that fails alone would hang the others' collectives
            also["configs4_per_gpu"] = {"failed": repr(e)}
        torch.cuda.empty_cache()
    if world > 1 and not args.no_extra:
        # what N GPUs of one node SHARE is only in these two: BASELINE configs[4] as a fixed total from host memory, and one file
        # in -> one file out over the ranks.  Every rank takes part (collectives inside); a failure ends the job on every rank.
        for name, fn in (("configs4_fixed_total", lambda: leg_fixed_total(dev, args, host_cores(), world, rank, barrier)),
                         ("sharded_file", lambda: leg_sharded_file(dev, args, world, rank, barrier))):
            note("leg " + name)
            r_ = fn()
            if rank == 0:
                also[name] = r_
            torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_extra:
            legs = (("configs1", lambda: leg_configs1(dev, args, host_cores())),
                    ("configs2", lambda: leg_configs2(dev, args, host_cores())),
                    ("ragged_lengths", lambda: leg_ragged(dev, args, host_cores(), out["read_bp_per_s"])),
                    ("ultralong", lambda: leg_ultralong(dev, args, host_cores())),
                    ("from_host_memory", lambda: leg_host_buffers(dev, args)),
                    ("end_to_end", lambda: leg_end_to_end(dev, args)),
                    ("end_to_end_gz", lambda: leg_end_to_end_gz(dev, args, host_cores())))
            for name, leg in legs:
                note("leg " + name)
                try:
                    also[name] = leg()
                except Exception as e:   # an extra leg must never break the bench line
                    also[name] = {"failed": repr(e)}
                torch.cuda.empty_cache()
            if "reads_per_s" in also.get("from_host_memory", {}):
                out["value_incl_h2d"] = also["from_host_memory"]["reads_per_s"]
        if also:
            out["config"]["also_measured"] = also
        full_path = args.full_json or os.path.join(REPO, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1)
        except Exception:
            pass
        print(json.dumps(compact_line(out), separators=(",", ":")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
