#!/usr/bin/env python3
"""GPU box: random adapter lists / thresholds / reads through the prefilter over the 2-bit plane (pc_prefilter_packed) and over
bytes (pc_prefilter_device), and -- adapters up to 32 bases -- the oracle's plain edit-distance DP: on reads made of A/C/G/T
the masks must be EQUAL; on reads with other letters the plane's mask must be a superset; no pair within the bound may be
dropped by either.   python tools/fuzz_prefilter_packed.py [rounds] [seed]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import porechop_amd
from oracle.oracle import Oracle
from porechop_amd.io import pack_reads
from tests.pairgen import mutate

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ora = Oracle()
dev = torch.device("cuda")
t0 = time.time()
tot_pairs = tot_equal = tot_super = refused = checked_dp = 0
for it in range(rounds):
    na = rng.choice([1, 2, 4, 7, 33, 90])
    ads = ["".join(rng.choice("ACGT") for _ in range(rng.choice([12, 16, 22, 24, 24, 28, 30, 32, 33, 40, 64]))) for _ in range(na)]
    thr = rng.choice([88.0, 90.0, 92.0, 95.0, 99.0])
    alphabet = rng.choice(["ACGT", "ACGT", "ACGTN", "ACGT-", "acgtACGTUu"])
    n = rng.choice([200, 1500])
    reads = []
    for _ in range(n):
        ln = rng.choice([0, 3, 40, 150, 700, 3000, 20000 if rng.random() < 0.05 else 900])
        r = [rng.choice(alphabet) for _ in range(ln)]
        if ln > 60 and rng.random() < 0.6:
            m = mutate(rng, rng.choice(ads), rng.choice([0.0, 0.04, 0.08, 0.12]))
            p = rng.randint(0, ln - 1)
            r[p:p + len(m)] = list(m)
            r = r[:ln]
        reads.append("".join(r).upper())
    arr = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.int64)]).astype(np.int64)
    d_off, d_len = torch.from_numpy(offs).to(dev), torch.from_numpy(lens).to(dev)
    pk, exc = pack_reads(arr)
    plane = torch.zeros(pk.size + 64, dtype=torch.uint8, device=dev)
    plane[:pk.size] = torch.from_numpy(pk).to(dev)
    arena = torch.from_numpy(np.concatenate([arr, np.full(64, ord("N"), np.uint8)])).to(dev)
    al = porechop_amd.Aligner(ads)
    al.set_length_hint(rng.choice([0, 0, 900]))
    ks = [al.max_edits(len(a), thr) for a in ads]
    ids = list(range(na))
    mb = al.prefilter_mask(arena, d_off, d_len, max(1, int(lens.max())), ids, ks)
    mp = al.prefilter_mask_packed(plane, d_off, d_len, max(1, int(lens.max())), ids, ks)
    al.sync()
    al.close()
    if mp is None:
        refused += 1
        continue
    a, b = mb.cpu().numpy().astype(np.uint32), mp.cpu().numpy().astype(np.uint32)
    tot_pairs += n * na
    only_acgt = set(alphabet.upper()) <= set("ACGTU")
    if only_acgt:
        assert np.array_equal(a, b), ("masks differ on A/C/G/T reads", it, thr, na)
        tot_equal += n * na
    else:
        assert np.all((a & ~b) == 0), ("the plane's mask dropped a pair the byte route keeps", it, thr, na)
        tot_super += n * na
    for j, (ad, k) in enumerate(zip(ads, ks)):
        if len(ad) <= 32 and na <= 7:
            d = ora.min_edits_many(arr, offs, lens, ad)
            bit = (b[:, j // 32] >> (j % 32)) & 1
            assert not ((d <= k) & (lens > 0) & (bit == 0)).any(), ("a pair within the bound was dropped", it, ad, k)
            if only_acgt:
                assert not ((d > k) & (bit == 1)).any(), ("a pair beyond the bound was kept (<= 32 bases: exact)", it, ad, k)
            checked_dp += n
print("prefilter over the plane vs over bytes: %d rounds, %d (window, adapter) pairs: %d equal (A/C/G/T reads), %d superset (other letters), "
      "%d pairs also against the plain DP, %d adapter lists refused by the packed route; 0 violations; %.0f s"
      % (rounds, tot_pairs, tot_equal, tot_super, checked_dp, refused, time.time() - t0))
