"""Reads that STAY at 2 bits per base (DeviceReads.packed_only): the host logic of the packed route -- end windows unpacked
for phases A / B, the prefilter over the plane, only the survivors (and the dirty reads) turned into bytes -- over the
oracle-backed stand-in (no GPU).  Trims and middle hits must equal the byte route's, with N runs, '-' masks and lower-case
bases in the reads, and with an adapter list the packed route refuses (an adapter holding an N)."""
import numpy as np
import torch

from porechop_amd.io import pack_reads
from porechop_amd.pipeline import AdapterSet, DeviceReads, Pipeline, ScanParams
from tests.cpu_aligner import OracleAligner
from tests.longgen import Y_BOTTOM, Y_TOP, make_read, mutated


def build(oracle, sets):
    p = ScanParams()
    return Pipeline(sets, p, aligner=OracleAligner(oracle, p.scores))


def reads_for_test():
    rng = np.random.default_rng(5)
    out = []
    for i in range(40):
        n = int(rng.integers(30, 4000))
        plants = []
        if i % 3 == 0 and n > 600:
            plants.append((n // 2, Y_BOTTOM + Y_TOP))
        if i % 7 == 0 and n > 900:
            plants.append((n - 200, mutated(rng, Y_TOP, 1, 0, 0)))
        body = make_read(n, 100 + i, plants, n_runs=[(n // 4, 9)] if i % 4 == 0 else (), dash_runs=[(n // 5, 5)] if i % 5 == 0 else ())
        if i % 6 == 0:
            body = body[:n // 3] + body[n // 3:2 * n // 3].lower() + body[2 * n // 3:]
        out.append((Y_TOP if i % 2 == 0 else "") + body + (Y_BOTTOM if i % 3 != 1 else ""))
    return out


def run_both(oracle, sets):
    reads = reads_for_test()
    blob = "".join(reads).encode()
    arena = np.frombuffer(blob + b"N" * 64, dtype=np.uint8).copy()
    lens = torch.tensor([len(r) for r in reads], dtype=torch.int32)
    off = torch.cumsum(lens.to(torch.int64), 0) - lens.to(torch.int64)
    res = []
    for packed in (False, True):
        pl = build(oracle, sets)
        if packed:
            pk, exc = pack_reads(arena, len(blob))
            dr = DeviceReads.packed_only(pl.aligner, torch.from_numpy(pk), len(blob), torch.from_numpy(exc), off, lens, end_size=pl.p.end_size)
            assert dr.arena is None
        else:
            dr = DeviceReads(torch.from_numpy(arena), off, lens)
        bs, be = pl.phase_a(dr)
        matching = pl.matching_sets(bs, be)
        st, et = pl.phase_b(dr, matching)[:2]
        h = pl.phase_c(dr, st, et, matching, prefilter=True)
        res.append((matching, st.tolist(), et.tolist(),
                    sorted(zip(h.read.tolist(), h.adapter.tolist(), h.start.tolist(), h.end.tolist())), dict(pl.stats), dr))
    return res


def test_packed_only_reads_give_the_byte_routes_results(oracle):
    sets = [AdapterSet("SQK-NSK007", ("SQK-NSK007_Y_Top", Y_TOP), ("SQK-NSK007_Y_Bottom", Y_BOTTOM))]
    (m0, st0, et0, h0, _, _), (m1, st1, et1, h1, stats, dr) = run_both(oracle, sets)
    assert (m0, st0, et0, h0) == (m1, st1, et1, h1)
    assert len(h0) >= 10 and sum(st0) > 0 and sum(et0) > 0
    assert dr.arena is None and stats.get("bases_unpacked_after_prefilter", 0) > 0 and "packed_route_refused" not in stats
    # only the survivors were turned into bytes
    assert stats["bases_unpacked_after_prefilter"] < 0.8 * dr.nbases


def test_adapter_list_the_packed_route_refuses_falls_back_to_bytes(oracle):
    with_n = Y_TOP[:10] + "N" + Y_TOP[11:]
    sets = [AdapterSet("SQK-NSK007", ("SQK-NSK007_Y_Top", Y_TOP), ("SQK-NSK007_Y_Bottom", Y_BOTTOM)),
            AdapterSet("with N", ("n_top", with_n), None)]
    (m0, st0, et0, h0, _, _), (m1, st1, et1, h1, stats, dr) = run_both(oracle, sets)
    assert (m0, st0, et0, h0) == (m1, st1, et1, h1)
    if len(m1) > 1:                                    # the N adapter matched too: its list cannot take the packed route
        assert stats.get("packed_route_refused", 0) >= 1 and dr.arena is not None
