"""Shared driver of the whole-run golden cases (tests/golden/runner_goldens.json, minted from the
reference CLI by tests/golden/make_golden.py on the seeded inputs of tests/readgen.py)."""
import json
import os

from tests import readgen

GOLDENS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "runner_goldens.json")

_FLAGS = {"--no_split": "no_split", "--discard_middle": "discard_middle", "--require_two_barcodes": "require_two_barcodes",
          "--discard_unassigned": "discard_unassigned", "--untrimmed": "untrimmed"}
_VALUES = {"--format": ("format", str), "--min_split_read_size": ("min_split_read_size", int),
           "--extra_middle_trim_good_side": ("extra_middle_trim_good_side", int),
           "--extra_middle_trim_bad_side": ("extra_middle_trim_bad_side", int), "--end_size": ("end_size", int),
           "--min_trim_size": ("min_trim_size", int), "--extra_end_trim": ("extra_end_trim", int),
           "--end_threshold": ("end_threshold", float), "--check_reads": ("check_reads", int),
           "--adapter_threshold": ("adapter_threshold", float), "--middle_threshold": ("middle_threshold", float),
           "--barcode_threshold": ("barcode_threshold", float), "--barcode_diff": ("barcode_diff", float),
           "--scoring_scheme": ("scoring_scheme", lambda v: tuple(int(x) for x in v.split(",")))}


# too much alignment work for the CPU stand-in (hundreds of adapters x many mask rounds): GPU test only
GPU_ONLY = {"native_loose"}


def load_cases():
    with open(GOLDENS) as f:
        return json.load(f)["cases"]


def options_from_argv(argv):
    from porechop_amd.runner import Options
    o = Options()
    it = iter(argv)
    for t in it:
        if t in _FLAGS:
            setattr(o, _FLAGS[t], True)
        else:
            attr, conv = _VALUES[t]
            setattr(o, attr, conv(next(it)))
    return o


def run_case(name, case, workdir, datasets, make_aligner=None, device=None):
    """Builds the input (cached per dataset in `datasets`), runs porechop_amd.runner, returns
    {output file -> md5 of content} to compare with case['outputs']."""
    from porechop_amd import runner
    if case["dataset"] not in datasets:
        path = readgen.build_dataset(case["dataset"], os.path.join(workdir, "datasets"))
        assert readgen.dataset_sha1(path) == case["input_sha1"], "tests/readgen.py drifted from the goldens"
        datasets[case["dataset"]] = path
    inp = datasets[case["dataset"]]
    opts = options_from_argv(case["argv"])
    work = os.path.join(workdir, "run_" + name)
    os.makedirs(work)
    kw = {"options": opts, "device": device}
    if make_aligner is not None:
        kw["aligner"] = make_aligner(opts.scoring_scheme)
    target = os.path.join(work, "bins" if case["mode"] == "b" else case["mode"][2:])
    try:
        if case["mode"] == "b":
            runner.run(inp, barcode_dir=target, **kw)
        else:
            runner.run(inp, output=target, **kw)
    except runner.UsageError as e:
        # the reference ends such runs with sys.exit(message): same message, no output
        assert case["exit"] == str(e), (name, case["exit"], str(e))
        return {}
    assert case["exit"] is None, (name, "the reference exits with: " + str(case["exit"]))
    if not os.path.exists(target):
        return {}                  # a rank other than 0 of a sharded run writes nothing
    return readgen.output_md5s(target)
