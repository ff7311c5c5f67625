"""World-size-2 run of the END-TO-END runner on CPU (gloo): reads sharded by bases, phase A's
presence table MAX-all-reduced, per-read results gathered in rank order, rank 0 writes -- the output
files must be the single-process ones, i.e. the reference CLI's (tests/golden/runner_goldens.json).
The oracle stands in for the GPU through tests/cpu_aligner.py (test infrastructure)."""
import os
import socket

import torch.multiprocessing as mp

CASES = ["native_default", "native_bins", "albacore_bins_check30", "ligation_default", "edge_default", "nothing_found"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, workdir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from tests.cpu_aligner import OracleAligner
    from tests.runner_cases import load_cases, run_case
    oracle = Oracle()
    cases = load_cases()
    datasets = {}
    out = {}
    for name in CASES:
        # both ranks build the same inputs (seeded) in their own directory and run the same command
        got = run_case(name, cases[name], os.path.join(workdir, "rank%d" % rank), datasets,
                       make_aligner=lambda sc: OracleAligner(oracle, sc))
        dist.barrier()
        out[name] = got
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_runner_outputs(tmp_path):
    from tests.runner_cases import load_cases
    cases = load_cases()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in CASES:
        assert got[0][name] == cases[name]["outputs"], (name, got[0][name])
        assert got[1][name] == {}, name          # only rank 0 writes
