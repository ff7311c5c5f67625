"""Oracle-driven mirror of the reference's per-read caller logic (TEST INFRASTRUCTURE).

Line-for-line restatement of what porechop/nanopore_read.py does with the 7-field strings:
  align_adapter            nanopore_read.py:476-491
  align_adapter_set        nanopore_read.py:149-164   (phase A)
  find_start_trim/end_trim nanopore_read.py:166-208   (phase B)
  find_middle_adapters     nanopore_read.py:210-243   (phase C)
driven by any `adapter_alignment(read, adapter, scores) -> str` callable (the CPU oracle in the
tests), so the batched GPU pipeline can be compared with the sequential reference semantics.
"""


def align_adapter(fn, read_seq, adapter_seq, scores):
    parts = fn(read_seq, adapter_seq, scores).split(",")
    read_start = int(parts[0])
    if read_start == -1:
        return 0.0, 0.0, -1, 0
    return float(parts[6]), float(parts[5]), read_start, int(parts[1]) + 1


def phase_a(fn, seqs, sets, p):
    best_start = [0.0] * len(sets)
    best_end = [0.0] * len(sets)
    for seq in seqs:
        for i, s in enumerate(sets):
            if "(full sequence)" in s.name:
                continue
            if s.start is not None:
                sc = align_adapter(fn, seq[:p.end_size], s.start[1], p.scores)[0]
                best_start[i] = max(best_start[i], sc)
            if s.end is not None:
                sc = align_adapter(fn, seq[-p.end_size:], s.end[1], p.scores)[0]
                best_end[i] = max(best_end[i], sc)
    return best_start, best_end


def phase_b(fn, seq, sets, matching, p):
    start_trim = end_trim = 0
    for i in matching:
        s = sets[i]
        if s.start is not None:
            full, partial, rs, re = align_adapter(fn, seq[:p.end_size], s.start[1], p.scores)
            if partial > p.end_threshold and re != p.end_size and re - rs >= p.min_trim_size:
                start_trim = max(start_trim, re + p.extra_end_trim)
    for i in matching:
        s = sets[i]
        if s.end is not None:
            full, partial, rs, re = align_adapter(fn, seq[-p.end_size:], s.end[1], p.scores)
            if partial > p.end_threshold and rs != 0 and re - rs >= p.min_trim_size:
                end_trim = max(end_trim, (p.end_size - rs) + p.extra_end_trim)
    return start_trim, end_trim


def trimmed(seq, start_trim, end_trim):
    if not start_trim and not end_trim:
        return seq
    return seq[start_trim:len(seq) - end_trim]


def phase_c(fn, seq, start_trim, end_trim, adapters, p):
    masked = trimmed(seq, start_trim, end_trim)
    hits = []
    for ai, (_, aseq) in enumerate(adapters):
        while True:
            full, _, rs, re = align_adapter(fn, masked, aseq, p.scores)
            if full >= p.middle_threshold:
                masked = masked[:rs] + "-" * (re - rs) + masked[re:]
                hits.append((ai, rs, re, full))
            else:
                break
    return hits
