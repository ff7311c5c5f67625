"""`python bench.py --gpus N` starts its own N ranks (task 2): the plain command line -- no torchrun around it --
must come up as N processes of one rendezvous, rank 0 printing the one line; under a launcher (WORLD_SIZE set) nothing is
re-launched.  PC_BENCH_LAUNCH_PROBE=1 stops every rank right after the process group is up, so this runs without a GPU
(gloo); the measuring run of the same plain command is tests/test_gpu_distributed.py."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, PC_BENCH_LAUNCH_PROBE="1", PC_DIST_BACKEND="gloo", **env)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=REPO, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE line, from rank 0
    return json.loads(lines[0])


def test_plain_command_with_gpus_2_starts_two_ranks():
    d = _run(["--gpus", "2", "--steps", "7"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo" and d["self_launched"] is True
    assert sorted(r[0] for r in d["ranks"]) == [0, 1] and sorted(r[1] for r in d["ranks"]) == [0, 1]
    assert len({r[2] for r in d["ranks"]}) == 2          # two processes
    assert d["steps"] == 7                               # the command line reaches the ranks unchanged


def test_plain_command_with_gpus_3():
    d = _run(["--gpus", "3"])
    assert d["n_gpus"] == 3 and sorted(r[0] for r in d["ranks"]) == [0, 1, 2]


def test_gpus_1_is_one_process_and_not_relaunched():
    d = _run(["--gpus", "1"])
    assert d["n_gpus"] == 1 and d["self_launched"] is False and len(d["ranks"]) == 1


def test_under_a_launcher_nothing_is_relaunched():
    """The driver's own line: torch.distributed.run ... bench.py --gpus 2."""
    e = dict(os.environ, PC_BENCH_LAUNCH_PROBE="1", PC_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", "bench.py", "--gpus", "2"], cwd=REPO, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["self_launched"] is False
