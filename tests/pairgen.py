"""Seeded generators of (read window, adapter) cases for the parity tests.

The shapes follow what Porechop actually sends through ``adapter_alignment``
(porechop/nanopore_read.py:149-243): 150-bp end windows (or whole short reads) against
22-50-bp panel adapters and 63-111-bp generated barcode adapters, plus whole reads for the
middle scan.  Edge cases mirror SURVEY.md section 8a: N's, '-' masks, lower case / U, reads
shorter than the adapter, truncated / mutated adapter copies overhanging either end.
"""
import random

SCHEMES = [
    (3, -6, -5, -2),    # Porechop default (porechop/porechop.py:145)
    (1, -1, -3, -1),
    (5, -4, -10, -1),
    (2, -3, -5, -2),
    (3, -6, -2, -5),    # gap_extend more negative than gap_open
    (1, -5, -1, -3),
]

# gap_open == gap_extend: the reference switches to its linear-gap recurrence
LINEAR_SCHEMES = [
    (3, -6, -5, -5),
    (2, -3, -4, -4),
    (1, -1, -1, -1),
    (5, -4, -2, -2),
]

# lengths only; sequences themselves are random (the real panel is exercised by the goldens)
ADAPTER_LENS = [1, 3, 8, 22, 24, 24, 28, 28, 32, 33, 40, 50, 63, 64, 65, 68, 102, 111]
READ_LENS = [1, 2, 5, 20, 50, 149, 150, 150, 150, 151, 300]


def mutate(rng, seq, rate=0.12):
    out = []
    for c in seq:
        x = rng.random()
        if x < rate * 0.4:
            out.append(rng.choice("ACGT"))          # substitution
        elif x < rate * 0.7:
            pass                                    # deletion
        elif x < rate:
            out.append(c)
            out.append(rng.choice("ACGT"))          # insertion
        else:
            out.append(c)
    return "".join(out)


def random_case(rng, n=None, m=None, alphabet=None):
    n = n if n is not None else rng.choice(READ_LENS)
    m = m if m is not None else rng.choice(ADAPTER_LENS)
    alphabet = alphabet or rng.choice(["ACGT", "ACGT", "ACGT", "ACGTN", "AC", "ACGT-", "acgtACGTUu"])
    ad = "".join(rng.choice("ACGT") for _ in range(m))
    if rng.random() < 0.05:
        ad = "".join(rng.choice("ACGTN") for _ in range(m))
    rd = "".join(rng.choice(alphabet) for _ in range(n))
    if rng.random() < 0.65 and n > 5:
        mut = mutate(rng, ad, rate=rng.choice([0.0, 0.05, 0.12, 0.25]))
        if mut and rng.random() < 0.3:
            k = rng.randint(0, len(mut) - 1)
            mut = mut[k:] if rng.random() < 0.5 else mut[:len(mut) - k]
        pos = rng.randint(0, n - 1)
        joined = rd[:pos] + mut + rd[pos:]
        rd = joined[:n] if rng.random() < 0.5 else joined[-n:]
        if rng.random() < 0.2:   # a second copy -> tie / earliest-hit situations
            pos = rng.randint(0, n - 1)
            joined = rd[:pos] + mut + rd[pos:]
            rd = joined[:n]
    if not rd:
        rd = "A"
    return rd, ad


def case_stream(seed, count, **kw):
    rng = random.Random(seed)
    for _ in range(count):
        yield random_case(rng, **kw)


def synthetic_read(rng, length=8000, start_adapter=None, end_adapter=None, chimera=None):
    """One synthetic read in the style of SURVEY.md section 8d (config 2/4)."""
    body = "".join(rng.choice("ACGT") for _ in range(length))
    if chimera is not None:
        pos = rng.randint(length // 8, 7 * length // 8)
        body = body[:pos] + mutate(rng, chimera, 0.05) + body[pos:]
    pre = ""
    if start_adapter is not None:
        pre = mutate(rng, start_adapter, 0.10)
        if rng.random() < 0.3:
            pre = pre[rng.randint(0, 20):]
    post = ""
    if end_adapter is not None:
        post = mutate(rng, end_adapter, 0.10)
        if rng.random() < 0.3:
            post = post[:max(0, len(post) - rng.randint(0, 20))]
    return pre + body + post
