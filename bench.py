#!/usr/bin/env python3
"""bench.py -- end + middle adapter scan of synthetic 8 kb reads on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over this rank's batch of reads, inputs already resident in
HBM: phase A (adapter-set presence over the check reads, 119-set panel, MAX all-reduce across
ranks -- the only collective), phase B (end windows vs the matching sets -> trim amounts) and
phase C (whole trimmed reads vs the matching sets' adapters, including the sequential
mask-and-realign rounds for reads with middle hits).  Workload = BASELINE.json configs[3]
("1M synthetic 8 kb reads with 1% chimeras, middle scan enabled") per GPU; reads are sharded
over ranks with no data-path collective (weak scaling).

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (score-only whole-read scan), timed with HIP events on its
                  launch stream inside the timed region (pc_get_timing)
  cpu_baseline -- the same phases B+C on a bounded sample of the same reads on the host cores,
                  through the compiled reference (oracle/_ref) when present, else the oracle port
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


def load_panel_sets():
    from porechop_amd.pipeline import AdapterSet
    with open(os.path.join(REPO, "tests", "golden", "panel.json")) as f:
        panel = json.load(f)
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None,
                       tuple(a["end"]) if a["end"] else None) for a in panel]


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


def one_step(pl, reads, n_check, world, proofs=False):
    """The hot path over one resident batch.  Returns (matching, start_trim, end_trim, hits).
    proofs=True is the optional variant with the two exact prunings of DESIGN.md section 7 (f-4 and
    the proven middle scan); the headline measurement never uses it."""
    from porechop_amd.distributed import reduce_presence
    check = None if n_check >= reads.n else torch.arange(n_check, device=reads.off.device)
    best_s, best_e = pl.phase_a(reads, check, prune=proofs)
    # adapter-set presence is the one cross-read reduction (porechop.py:327): 119 x 2 maxima, MAX
    # all-reduce over RCCL (a no-op at world size 1)
    best_s, best_e = reduce_presence(best_s, best_e)
    matching = pl.matching_sets(best_s, best_e)
    st, et = pl.phase_b(reads, matching)
    hits = pl.phase_c(reads, st, et, matching, prove=proofs)
    return matching, st, et, hits


def cpu_baseline(reads, pl, matching, seconds, workers):
    """Phases B + C of the same reads, the reference's sequential per-read logic
    (tests/ref_pipeline.py), on all host cores: one spawned worker PROCESS per core (Porechop's own
    --threads pool is GIL-bound, README.md:355-359; processes are its fair upper bound).
    Bounded: the sample is sized from a probe so the leg takes ~`seconds`."""
    import multiprocessing as mp
    from dataclasses import asdict
    from oracle.oracle import REF_SO
    from tests.cpu_worker import run_chunk

    kind = "reference" if os.path.isfile(REF_SO) else "port"
    n_pull = min(reads.n, 16384)
    ln = int(reads.length[0].item())
    host = reads.arena[: n_pull * ln].cpu().numpy().tobytes().decode("ascii")
    seqs = [host[i * ln:(i + 1) * ln] for i in range(n_pull)]
    sets = [(s.name, s.start, s.end) for s in pl.sets]
    params = asdict(pl.p)
    _, t_probe = run_chunk((seqs[:4], sets, matching, params, True))
    per_read = t_probe / 4
    n = int(min(n_pull, max(workers * 2, seconds * workers / max(per_read, 1e-6))))
    per = max(1, n // workers)
    chunks = [seqs[i:i + per] for i in range(0, n, per)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        pool.map(run_chunk, [(c[:1], sets, matching, params, True) for c in chunks])   # start-up + import, untimed
        t0 = time.perf_counter()
        res = pool.map(run_chunk, [(c, sets, matching, params, True) for c in chunks])
        dt = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    return {"value": done / dt, "unit": "reads/s", "cores": workers, "kind": kind,
            "sample": "%d of the benchmark's reads (%d bp each), phases B+C, %d worker processes over the %s, %.1f s wall"
                      % (done, ln, workers, "compiled reference oracle/_ref/cpp_functions.so" if kind == "reference"
                         else "oracle port oracle/pc_oracle.c", dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=8000)
    ap.add_argument("--chimera", type=float, default=0.01)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    args = ap.parse_args()

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
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from porechop_amd.pipeline import Pipeline, ScanParams
    from porechop_amd.synth import make_reads

    params = ScanParams()
    pl = Pipeline(load_panel_sets(), params, device=dev)
    # seed 3 = BASELINE config 4; every rank draws its own shard (rank-dependent seed)
    reads = make_reads(args.reads, args.read_len, seed=3 + 1000 * rank, start_frac=0.9, end_frac=0.5,
                       chimera_frac=args.chimera, device=dev)
    n_check = max(1, params.check_reads // world)   # each rank checks its share of the first 10 000

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    matching = None
    for _ in range(args.warmup):
        matching, st, et, hits = one_step(pl, reads, n_check, world)
        pl.aligner.sync()
    pl.aligner.set_timing(True)
    pl.stats = {k: 0 for k in pl.stats}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        matching, st, et, hits = one_step(pl, reads, n_check, world)
    pl.aligner.sync()
    barrier()
    dt = time.perf_counter() - t0
    timing = pl.aligner.get_timing()
    pl.aligner.set_timing(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- AFTER the headline measurement: the same steps with the two optional exact prunings
    # (score bound for the identity thresholds: fewer tracebacks, identical sets / trims / hits).
    # Reported as an extra field; `value` above never includes it.
    one_step(pl, reads, n_check, world, proofs=True)
    pl.aligner.sync()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        _, _, _, hits_p = one_step(pl, reads, n_check, world, proofs=True)
    pl.aligner.sync()
    barrier()
    dtp = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dtp, op=dist.ReduceOp.MAX)
    dt_proofs = float(dtp.item())
    same_hits = bool(hits_p.read.numel() == hits.read.numel() and torch.equal(hits_p.read, hits.read) and
                     torch.equal(hits_p.start, hits.start) and torch.equal(hits_p.end, hits.end))

    if rank == 0:
        total_reads = args.reads * world
        reads_per_s = total_reads * args.steps / dt
        # ---- roofline of the dominant kernel: the score-only whole-read scan ---------------
        # algorithmic bytes: per read |H| input bytes ONCE for all A middle
        # adapters + 28 B of result per (read, adapter); a launch scanning one of A adapters is
        # credited 1/A of the read bytes.  cells = sum |H| x |V|.
        # Dominant kernel = the run-time specialised scan (timing kind 'score_spec'); the generic
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
            # VALU ceiling measured with tools/ubench_valu.hip: one wave64 packed 16-bit op per ~4.3 cycles
            # per SIMD = 39.3 T lane-ops/s; the specialised score kernel spends 5 packed-fp16 ops per 2
            # cells (6 in its int16 variant; generic kernel: 9), so its ceiling is 39.3e12 * 2 / 5
            # cell updates per second.
            ops_per_pair = (6 if os.environ.get("PC_JIT_INT16", "0") not in ("", "0") else 5) if jit else 9
            valu_peak_gcups = 39.3e12 * 2 / ops_per_pair / 1e9
            roof = {"bound": "hbm",
                    "kernel": ("pc_spec_score (run-time specialised score-only whole-read scan)" if jit
                               else "scan_kernel<R,PAD,false> (generic score-only whole-read scan)"),
                    "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": None, "launches": int(launches), "avg_launch_ms": per_launch_s * 1e3,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "gcups": cells / per_launch_s / 1e9,
                    "valu": {"achieved_gcups": cells / per_launch_s / 1e9, "peak_gcups": valu_peak_gcups,
                             "frac": cells / per_launch_s / 1e9 / valu_peak_gcups, "ops_per_2_cells": ops_per_pair},
                    "note": "integer max-plus DP, >= 50 cells per algorithmic byte: VALU-bound by construction; "
                            "the launch average includes the small mask-and-realign launches of the same kernel "
                            "(DESIGN.md section 4)"}
            # HBM-side traffic of that kernel comes from the separately collected rocprofv3 --pmc passes
            # of this same command (tools/profile_round.sh -> profiles/<round>_summary.json): FETCH_SIZE +
            # WRITE_SIZE of one launch, KB as reported.  (The guide's x2 correction is for wide 16-B/lane
            # streams; here lanes gather 4 B each and the kernel provably consumes 7.95 GB per launch
            # against a reported FETCH_SIZE of 6.4 GB, so no doubling is applied.)
            try:
                import glob
                summ = sorted(glob.glob(os.path.join(REPO, "profiles", "*_summary.json")))
                if summ:
                    with open(summ[-1]) as f:
                        sj = json.load(f)
                    kk = sj["kernels"].get("pc_spec_score" if jit else "")
                    if kk and sj.get("reads_per_gpu") == args.reads and world == 1:
                        roof["traffic"] = (kk["FETCH_SIZE_KB_mean_launch"] + kk["WRITE_SIZE_KB_mean_launch"]) * 1024.0
                        roof["traffic_source"] = os.path.relpath(summ[-1], REPO) + " (mean over this kernel's launches; FETCH_SIZE + WRITE_SIZE)"
            except Exception:
                pass
        kern_ms = {k: v[0] / args.steps for k, v in timing.items()}
        out = {
            "metric": "reads/sec (and read-bp/sec) end+middle adapter scan, 8 kb reads",
            "value": reads_per_s, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16-held integers (score scan) / i16 (traced scan), exact", "data": "synthetic",
            "read_bp_per_s": reads_per_s * args.read_len,
            "config": {"workload": "BASELINE configs[3]: %d synthetic %d-bp reads per GPU, %.0f%% chimeras, "
                                   "phases A (119-set panel, %d check reads) + B + C (middle scan on)"
                                   % (args.reads, args.read_len, args.chimera * 100, params.check_reads),
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "parallelism": "reads sharded x%d" % world,
                       "matching_sets": [pl.sets[i].name for i in matching],
                       "middle_hits_per_step": int(hits.read.numel()), "mask_rounds": hits.rounds,
                       "kernel_ms_per_step": kern_ms,
                       "optional_exact_pruning": {"reads_per_s": total_reads * args.steps / dt_proofs,
                                                  "ms_per_step": dt_proofs / args.steps * 1e3, "same_middle_hits": same_hits,
                                                  "note": "not the headline: phase A and the middle scan trace back only "
                                                          "pairs whose score can still reach the identity threshold "
                                                          "(DESIGN.md section 7)"}},
            "roofline": roof,
        }
        if args.cpu_seconds > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(reads, pl, matching, args.cpu_seconds, host_cores())
                out["config"]["speedup_vs_cpu_baseline"] = reads_per_s / out["cpu_baseline"]["value"]
            except Exception as e:   # the baseline leg must never break the bench line
                out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
