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


# ---- barcode calling (demultiplexing) ------------------------------------------------------------
def is_barcode(s):
    return s.name.startswith("Barcode ")                      # adapters.py:32-33


def barcode_direction(s):
    return "reverse" if "_rev" in s.start[0] else "forward"  # adapters.py:35-39


def barcode_name(s):
    names = [s.name]                                          # adapters.py:41-52
    if s.start is not None:
        names.append(s.start[0])
    if s.end is not None:
        names.append(s.end[0])
    return sorted(names, key=lambda x: len(x))[0].replace(" ", "_")


def phase_b_barcodes(fn, seq, sets, matching, p, orientation):
    """find_start_trim + find_end_trim with check_barcodes=True (nanopore_read.py:166-208)
    -> start_trim, end_trim, start_barcode_scores, end_barcode_scores (dicts in insertion order)."""
    start_trim = end_trim = 0
    start_scores, end_scores = {}, {}
    for i in matching:
        s = sets[i]
        if s.start is None:
            continue
        full, partial, rs, re = align_adapter(fn, seq[:p.end_size], s.start[1], p.scores)
        if partial > p.end_threshold and re != p.end_size and re - rs >= p.min_trim_size:
            start_trim = max(start_trim, re + p.extra_end_trim)
        if is_barcode(s) and barcode_direction(s) == orientation:
            start_scores[barcode_name(s)] = full
    for i in matching:
        s = sets[i]
        if s.end is None:
            continue
        full, partial, rs, re = align_adapter(fn, seq[-p.end_size:], s.end[1], p.scores)
        if partial > p.end_threshold and rs != 0 and re - rs >= p.min_trim_size:
            end_trim = max(end_trim, (p.end_size - rs) + p.extra_end_trim)
        if is_barcode(s) and barcode_direction(s) == orientation:
            end_scores[barcode_name(s)] = full
    return start_trim, end_trim, start_scores, end_scores


def determine_barcode(start_scores, end_scores, barcode_threshold, barcode_diff, require_two_barcodes):
    """nanopore_read.py:399-466 (without the Albacore agreement rule) -> bin name."""
    sb = sorted(start_scores.items(), reverse=True, key=lambda x: x[1])
    eb = sorted(end_scores.items(), reverse=True, key=lambda x: x[1])
    best_s = sb[0] if len(sb) >= 1 else ("none", 0.0)
    second_s = sb[1] if len(sb) >= 2 else ("none", 0.0)
    best_e = eb[0] if len(eb) >= 1 else ("none", 0.0)
    second_e = eb[1] if len(eb) >= 2 else ("none", 0.0)
    if require_two_barcodes:
        ok = (best_s[1] >= barcode_threshold and best_e[1] >= barcode_threshold and
              best_s[1] >= second_s[1] + barcode_diff and best_e[1] >= second_e[1] + barcode_diff and
              best_s[0] == best_e[0])
        return best_s[0] if ok else "none"
    allb, seen = [], set()
    for name, score in sorted(sb + eb, reverse=True, key=lambda x: x[1]):
        if name not in seen:
            allb.append((name, score))
            seen.add(name)
    best = allb[0] if len(allb) >= 1 else ("none", 0.0)
    second = allb[1] if len(allb) >= 2 else ("none", 0.0)
    if best[1] >= barcode_threshold and best[1] >= second[1] + barcode_diff:
        return best[0]
    return "none"
