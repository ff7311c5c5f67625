#!/usr/bin/env python3
"""Run the reference's OWN, UNCHANGED `porechop.porechop.main()` (what porechop-runner.py calls) from the staged copy
under oracle/_ref/porechop_ref (made by `make -C oracle ref`; git-ignored, it travels to the GPU box) with a stopwatch
around each phase driver -- TEST / BENCH INFRASTRUCTURE, never imported by porechop_amd/.

    python tests/ref_cli.py [--stage DIR] [--dropin] [--report OUT.json] -- -i reads.fastq -o out.fastq --threads 16 -v 0

  (default)   the reference exactly as shipped: its Python over its own compiled cpp_functions.so -- BASELINE.md's B1,
              "Porechop's own --threads CPU path" (porechop/porechop.py:86,108,484-509,575-591).
  --stage DIR another staged tree, e.g. one whose porechop/cpp_functions.so is libporechop_amd.so (INTEGRATION.md mode A:
              only the shared object swapped).
  --dropin    INTEGRATION.md mode B: `porechop_amd.dropin.install(pp)` before `pp.main()` -- the three phase drivers batch
              what they are about to ask onto the GPU, then run unchanged over the memo.

The report holds the wall clock of load_reads / find_matching_adapter_sets / find_adapters_at_read_ends /
find_adapters_in_read_middles / output_reads (porechop.py:34-78), the whole of main(), and with --dropin the memo's
statistics (misses must be 0)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_STAGE = os.path.join(REPO, "oracle", "_ref", "porechop_ref")
PHASES = ("load_reads", "find_matching_adapter_sets", "find_adapters_at_read_ends", "find_adapters_in_read_middles",
          "output_reads")


def staged(stage=DEFAULT_STAGE):
    return os.path.isfile(os.path.join(stage, "porechop", "porechop.py")) and \
        os.path.isfile(os.path.join(stage, "porechop", "cpp_functions.so"))


def main():
    argv = sys.argv[1:]
    stage, dropin, report = DEFAULT_STAGE, False, None
    while argv and argv[0] != "--":
        a = argv.pop(0)
        if a == "--stage":
            stage = os.path.abspath(argv.pop(0))
        elif a == "--dropin":
            dropin = True
        elif a == "--report":
            report = argv.pop(0)
        else:
            sys.exit("ref_cli.py: unknown option %r" % a)
    argv = argv[1:]
    if not staged(stage):
        sys.exit("ref_cli.py: no staged reference under %s (run `make -C oracle ref` where /root/reference exists)" % stage)
    sys.path.insert(0, stage)
    import porechop.porechop as pp
    assert os.path.realpath(pp.__file__).startswith(os.path.realpath(stage)), pp.__file__
    state = None
    if dropin:
        if REPO not in sys.path:
            sys.path.insert(1, REPO)
        import porechop_amd.dropin as dropin_mod
        state = dropin_mod.install(pp)
    seconds = {}

    def stopwatch(name):
        fn = getattr(pp, name)

        def timed(*a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                seconds[name] = seconds.get(name, 0.0) + time.perf_counter() - t0
        return timed
    for name in PHASES:
        setattr(pp, name, stopwatch(name))
    sys.argv = ["porechop"] + argv
    t0 = time.perf_counter()
    pp.main()
    total = time.perf_counter() - t0
    out = {"main_s": total, "phase_s": seconds, "argv": argv, "stage": stage,
           "cpp_functions_so": os.path.realpath(os.path.join(stage, "porechop", "cpp_functions.so"))}
    if state is not None:
        out["dropin"] = dropin_mod.stats()
        out["dropin"]["prefetch_s"] = state.prefetch_seconds
        out["dropin"]["backend_s"] = state.backend_seconds
    if report:
        with open(report, "w") as f:
            json.dump(out, f)
    else:
        print(json.dumps(out), file=sys.stderr)


if __name__ == "__main__":
    main()
