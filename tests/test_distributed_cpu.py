"""World-size-2 run of the sharded path on CPU (gloo): each rank computes phase A on its shard of
the reads (the oracle stands in for the GPU here -- test infrastructure), the presence table is
MAX-all-reduced exactly as bench.py does over RCCL, and both ranks must arrive at the single-process
table and matching sets; per-read results gathered in rank order must equal the unsharded order."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.pairgen import synthetic_read

Y_TOP = "AATGTACTTCGTTCAGTTACGTATTGCT"
Y_BOTTOM = "GCAATACGTAACTGAACGAAGT"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs():
    from porechop_amd.pipeline import AdapterSet
    rng = random.Random(77)
    reads = [synthetic_read(rng, rng.choice([300, 500]), Y_TOP if i % 3 else None, Y_BOTTOM if i % 2 else None)
             for i in range(14)]
    # only rank 1's shard contains a clean copy of the third set's adapter
    reads[11] = "GGTTGTTTCTGTTGGTGCTGATATTGC" + reads[11]
    sets = [AdapterSet("SQK-NSK007", ("Y_Top", Y_TOP), ("Y_Bottom", Y_BOTTOM)),
            AdapterSet("Rapid", ("Rapid_adapter", "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"), None),
            AdapterSet("PCR", ("PCR_1_start", "GGTTGTTTCTGTTGGTGCTGATATTGC"), ("PCR_1_end", "GCAATATCAGCACCAACAGAAA"))]
    return reads, sets


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    from porechop_amd.distributed import gather_in_order, reduce_presence, shard_bounds
    from porechop_amd.pipeline import ScanParams
    from tests import ref_pipeline
    reads, sets = _make_inputs()
    p = ScanParams()
    lo, hi = shard_bounds(len(reads), world, rank)
    bs, be = ref_pipeline.phase_a(Oracle().adapter_alignment, reads[lo:hi], sets, p)
    bs, be = reduce_presence(torch.tensor(bs, dtype=torch.float64), torch.tensor(be, dtype=torch.float64))
    local = torch.arange(lo, hi, dtype=torch.int64)[:, None].repeat(1, 2)
    gathered = gather_in_order(local)
    q.put((rank, bs.tolist(), be.tolist(), gathered[:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_presence_reduction_and_order(oracle):
    from porechop_amd.distributed import shard_bounds
    from porechop_amd.pipeline import ScanParams
    from tests import ref_pipeline
    reads, sets = _make_inputs()
    want_bs, want_be = ref_pipeline.phase_a(oracle.adapter_alignment, reads, sets, ScanParams())
    # the third set is only detectable from rank 1's shard: the reduction has to carry it over
    lo1, hi1 = shard_bounds(len(reads), 2, 0)
    bs0, _ = ref_pipeline.phase_a(oracle.adapter_alignment, reads[lo1:hi1], sets, ScanParams())
    assert want_bs[2] >= 90.0 > bs0[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, bs, be, order in got:
        assert bs == want_bs and be == want_be, rank
        assert order == list(range(len(reads))), rank


def test_shard_bounds_cover_everything_once():
    from porechop_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            cuts = [shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
