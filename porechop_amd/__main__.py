"""`python -m porechop_amd -i reads.fastq -o trimmed.fastq` -- the option names of the reference's
command line (porechop/porechop.py:85-185) over porechop_amd.runner.run.  Progress tables are not
reproduced; -v 1 prints a short summary."""
import argparse
import sys

from .runner import Options, UsageError, run


def main(argv=None):
    d = Options()
    p = argparse.ArgumentParser(prog="porechop_amd", description="MI355X adapter trimming with Porechop's semantics")
    p.add_argument("-i", "--input", required=True)
    p.add_argument("-o", "--output")
    p.add_argument("--format", choices=["auto", "fasta", "fastq", "fasta.gz", "fastq.gz"], default=d.format)
    p.add_argument("-v", "--verbosity", type=int, default=1)
    p.add_argument("-t", "--threads", type=int, default=1, help="accepted for compatibility; alignment runs on the GPU")
    p.add_argument("-b", "--barcode_dir")
    p.add_argument("--barcode_threshold", type=float, default=d.barcode_threshold)
    p.add_argument("--barcode_diff", type=float, default=d.barcode_diff)
    p.add_argument("--require_two_barcodes", action="store_true")
    p.add_argument("--untrimmed", action="store_true")
    p.add_argument("--discard_unassigned", action="store_true")
    p.add_argument("--adapter_threshold", type=float, default=d.adapter_threshold)
    p.add_argument("--check_reads", type=int, default=d.check_reads)
    p.add_argument("--scoring_scheme", type=str, default=",".join(str(x) for x in d.scoring_scheme))
    p.add_argument("--end_size", type=int, default=d.end_size)
    p.add_argument("--min_trim_size", type=int, default=d.min_trim_size)
    p.add_argument("--extra_end_trim", type=int, default=d.extra_end_trim)
    p.add_argument("--end_threshold", type=float, default=d.end_threshold)
    p.add_argument("--no_split", action="store_true")
    p.add_argument("--discard_middle", action="store_true")
    p.add_argument("--middle_threshold", type=float, default=d.middle_threshold)
    p.add_argument("--extra_middle_trim_good_side", type=int, default=d.extra_middle_trim_good_side)
    p.add_argument("--extra_middle_trim_bad_side", type=int, default=d.extra_middle_trim_bad_side)
    p.add_argument("--min_split_read_size", type=int, default=d.min_split_read_size)
    # porechop.py:182-183 prints its bare version number and wrappers compare the whole line: exactly that, nothing after it
    # (the build string of this package is pc_version() / `python -c "import porechop_amd; print(porechop_amd.load_library().pc_version())"`)
    p.add_argument("--version", action="version", version="0.2.4")
    a = p.parse_args(argv)
    try:
        scheme = tuple(int(x) for x in a.scoring_scheme.split(","))
    except ValueError:
        sys.exit("Error: incorrectly formatted scoring scheme")
    if len(scheme) != 4:
        sys.exit("Error: incorrectly formatted scoring scheme")
    if a.threads < 1:
        sys.exit("Error: at least one thread required")
    opts = Options(**{k: getattr(a, k) for k in Options.__dataclass_fields__ if k != "scoring_scheme"}, scoring_scheme=scheme)
    # launched by `python -m torch.distributed.run --nproc-per-node N -m porechop_amd ...`: one process per GPU,
    # reads sharded over the ranks (runner.run), RCCL for the one small reduction
    import os
    device = None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = "cuda:%d" % local
        dist.init_process_group(os.environ.get("PC_DIST_BACKEND", "nccl"))
    try:
        res = run(a.input, output=a.output, barcode_dir=a.barcode_dir, options=opts, device=device)
    except (UsageError, ValueError) as e:
        sys.exit(str(e))
    except RuntimeError as e:                 # no GPU / no HIP library: there is no CPU path to fall back to
        sys.exit("Error: " + str(e))
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    if a.verbosity > 0:
        dest = sys.stderr if (a.output is None and a.barcode_dir is None) else sys.stdout
        print("%d reads; adapter sets: %s" % (res.n_reads, ", ".join(res.matching_sets) or "none"), file=dest)
        print("start-trimmed %d, end-trimmed %d, reads with middle adapters %d" %
              (res.counts.get("start_trimmed", int((res.start_trim > 0).sum())),
               res.counts.get("end_trimmed", int((res.end_trim > 0).sum())), res.middle_hit_reads), file=dest)
        for path, (n, bases) in sorted(res.files.items()):
            print("  %s: %d reads, %d bases" % (path, n, bases), file=dest)
        print("  " + ", ".join("%s %.2fs" % kv for kv in res.seconds.items()), file=dest)


if __name__ == "__main__":
    main()
