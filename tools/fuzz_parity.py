#!/usr/bin/env python3
"""Large randomised parity run (GPU box): the HIP library against the C oracle, bit for bit, on
millions of end-window pairs and tens of thousands of whole reads, over random valid scoring
schemes and adapters of 1..120 bases.  Prints one line per block and a final tally.
    python tools/fuzz_parity.py [blocks] [seed]
PC_FUZZ_UNRESTRICTED=1 draws ANY four integers as the scheme and adapters up to 400 bases (the plain-int32 kernel's ground).
With PC_CHECK_RANGE=1 PC_JIT_CHECK_RANGE=1 PC_JIT_MIN_CELLS=1 in the environment the range-checking builds of the 16-bit
kernels run (every row class of the packed-fp16 traced kernel, the specialised score kernel of every adapter pair met):
each block then also prints the extremes of every value those kernels formed (pc_debug_value_range; the exactness argument
needs |value| <= 2040 in the fp16 kernels, <= 32000 in the int16 ones -- a launch outside fails by itself)."""
import os
import random
import sys
import time

sys.path.insert(0, ".")
import numpy as np

import porechop_amd
from oracle.oracle import Oracle
from tests.pairgen import mutate

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
nrng = np.random.default_rng(seed)
o = Oracle()
bad = total = 0
checking = os.environ.get("PC_CHECK_RANGE", "0") not in ("", "0") or os.environ.get("PC_JIT_CHECK_RANGE", "0") not in ("", "0")
worst = [0, 0]
t0 = time.time()
unrestricted = os.environ.get("PC_FUZZ_UNRESTRICTED", "0") not in ("", "0")
for blk in range(blocks):
    while unrestricted:
        # ANY four integers, as the reference takes them (porechop.py:145,196-202): positive / zero gap scores, match <=
        # mismatch, magnitudes far beyond 16 bits; adapters up to 400 bases.  These run the plain-int32 kernel
        # (csrc/pc_slow.hip) wherever the packed kernels refuse; smaller blocks, it is ~100x slower per cell.
        kind = blk % 4
        if kind == 0:
            sc = tuple(rng.randint(-12, 12) for _ in range(4))
        elif kind == 1:
            sc = tuple(rng.randint(-100000, 100000) for _ in range(4))
        elif kind == 2:
            sc = (rng.randint(1, 8), rng.randint(-8, 0), rng.randint(0, 6), rng.randint(0, 6))
        else:
            sc = (rng.randint(1, 30), -rng.randint(0, 40), -rng.randint(1, 40), -rng.randint(1, 40))
        ads = ["".join(rng.choice("ACGT") for _ in range(rng.choice([1, 5, 22, 24, 28, 33, 64, 120, 129, 200, 400]))) for _ in range(6)]
        al = porechop_amd.Aligner(ads, scores=sc)
        break
    while not unrestricted:
        sc = (rng.randint(1, 30), -rng.randint(0, 40), -rng.randint(1, 40), -rng.randint(1, 40))
        if blk % 5 == 4:
            # match - mismatch of 1 or 2: the match count pc_walk.h derives from the score then divides by almost nothing,
            # so a wrong end-cell score would pass its divisibility check -- the range-checking builds count the matches
            # from the bases as well (traceback_pairs<COUNT>) and these schemes are where that matters
            mt = rng.randint(1, 6)
            sc = (mt, mt - rng.choice([1, 2]), sc[2], sc[3])
        if sc[0] > sc[1]:
            try:
                ads = ["".join(rng.choice("ACGT") for _ in range(rng.choice([1, 5, 22, 24, 24, 28, 33, 38, 50, 64, 87, 120]))) for _ in range(6)]
                al = porechop_amd.Aligner(ads, scores=sc)
                break
            except RuntimeError:
                continue
    whole = blk % 4 == 3
    n = 2000 if whole else 100_000
    if unrestricted:
        whole = blk % 8 == 7
        n = 300 if whole else 20_000
    lens = nrng.choice([3000, 8000] if not unrestricted else [900, 2500], size=n) if whole else nrng.choice([1, 7, 60, 149, 150, 150, 150, 151], size=n)
    lens = lens.astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.int64)
    alphabet = np.frombuffer(rng.choice([b"ACGT", b"ACGT", b"ACGTN", b"AC", b"ACGT-", b"acgtACGTUu"]), dtype=np.uint8)
    arena = alphabet[nrng.integers(0, len(alphabet), int(lens.sum()) + 64)]
    aidx = nrng.integers(0, len(ads), n).astype(np.int32)
    # implant mutated adapter copies in two thirds of the windows
    for i in nrng.choice(n, size=2 * n // 3, replace=False):
        m = mutate(rng, ads[aidx[i]], rng.choice([0.0, 0.05, 0.12, 0.25])).encode()
        if not m or lens[i] < 8:
            continue
        p = rng.randint(0, int(lens[i]) - 1)
        m = m[: int(lens[i]) - p]
        arena[offs[i] + p: offs[i] + p + len(m)] = np.frombuffer(m, dtype=np.uint8)
    only = os.environ.get("PC_FUZZ_ONLY")
    if only is not None and int(only) != blk:          # (the block's data is still drawn, so that the random streams stay in step)
        al.close()
        continue
    try:
        got = al.align_host(arena, offs, lens, aidx, porechop_amd.MODE_TWO_PASS if whole else porechop_amd.MODE_AUTO)
    except RuntimeError as e:
        print("block %2d scheme %-20s adapters %s FAILED: %s" % (blk, sc, [len(a) for a in ads], e), flush=True)
        bad += n; total += n
        al.close()
        continue
    rng_txt = ""
    if checking:
        lo, hi = al.debug_value_range()
        worst = [min(worst[0], lo), max(worst[1], hi)]
        rng_txt = "  values formed in [%d, %d]" % (lo, hi)
    ad_arena = np.frombuffer("".join(ads).encode(), dtype=np.uint8)
    ad_len = np.array([len(a) for a in ads], dtype=np.int32)
    ad_off = np.concatenate([[0], np.cumsum(ad_len[:-1].astype(np.int64))]).astype(np.int64)
    want = o.align_many(arena, offs, lens, ad_arena, ad_off[aidx], ad_len[aidx], sc)
    want8 = np.concatenate([want[:, :7], want[:, 8:9]], axis=1)
    ok = (got == want8).all(axis=1) | ((want[:, 0] == -1) & (got[:, 0] == -1))
    bad += int((~ok).sum())
    total += n
    print("block %2d scheme %-20s %s n=%6d mismatches=%d%s  (%.0f s)" % (blk, sc, "whole reads" if whole else "end windows", n, int((~ok).sum()), rng_txt, time.time() - t0), flush=True)
    if (~ok).any():
        i = int(np.nonzero(~ok)[0][0])
        print("   first:", arena[offs[i]:offs[i] + lens[i]].tobytes()[:200], ads[aidx[i]], got[i], want8[i])
    al.close()
print("TOTAL pairs=%d mismatches=%d" % (total, bad) + ("  extremes of all values formed by the range-checking kernels: [%d, %d]" % tuple(worst) if checking else ""))
