"""The adapter panel and the set-level rules that depend only on names and sequences.

The 119 sets (names + sequences) come from porechop_amd/panel.json, the copy recorded from
porechop/adapters.py:78-463 by tests/golden/make_golden.py; reading the installed reference module instead is an
explicit opt-in (load_panel).  The rules mirrored here:

  adapters.py:29-52    best_start_or_end_score, is_barcode, barcode_direction, get_barcode_name
  adapters.py:466-499  the three "full sequence" barcode adapters (flanking sequences are ONT's)
  porechop.py:374-390  fix_up_1d2_sets
  porechop.py:330-371  choose_barcoding_kit
  porechop.py:410-436  add_full_barcode_adapter_sets
"""
import json
import os
from typing import List, Optional

from .pipeline import AdapterSet

_HERE = os.path.dirname(os.path.abspath(__file__))


def panel_from_reference_module():
    """The panel as the UNCHANGED reference module defines it, when Porechop is importable in this
    environment (`porechop.adapters.ADAPTERS`, porechop/adapters.py:77-463) -- read at run time, so a
    Porechop with a newer adapter list is followed without touching this package.  None if it is not."""
    try:
        from porechop.adapters import ADAPTERS          # noqa: the reference's own table
    except Exception:
        return None
    out = []
    for a in ADAPTERS:
        st = tuple(a.start_sequence) if getattr(a, "start_sequence", None) else None
        en = tuple(a.end_sequence) if getattr(a, "end_sequence", None) else None
        out.append(AdapterSet(a.name, st, en))
    return out or None


PANEL_SOURCE = None          # what the last load_panel() call used: "recorded" or "reference module"


def load_panel(prefer_reference: Optional[bool] = None) -> List[AdapterSet]:
    """The adapter panel.  Default: the copy recorded from the reference into panel.json
    (tests/golden/make_golden.py) -- the panel every golden, test and benchmark of this package uses, so a run's
    output never depends on what happens to be importable under the name `porechop`.

    prefer_reference=True, or PORECHOP_AMD_PANEL=reference in the environment when the argument is None, is the
    explicit opt-in to read `porechop.adapters.ADAPTERS` from an installed Porechop at run time instead (a newer
    adapter list is then followed without touching this package; that module's code runs in this process).
    The source used is kept in PANEL_SOURCE and said on stderr when it is not the recorded panel."""
    global PANEL_SOURCE
    if prefer_reference is None:
        prefer_reference = os.environ.get("PORECHOP_AMD_PANEL", "").lower() == "reference"
    if prefer_reference:
        p = panel_from_reference_module()
        if p is not None:
            PANEL_SOURCE = "reference module"
            import sys
            print("porechop_amd: adapter panel read from the installed porechop.adapters module (%d sets)" % len(p),
                  file=sys.stderr)
            return p
    with open(os.path.join(_HERE, "panel.json")) as f:
        raw = json.load(f)
    PANEL_SOURCE = "recorded"
    return [AdapterSet(a["name"], tuple(a["start"]) if a["start"] else None, tuple(a["end"]) if a["end"] else None)
            for a in raw]


def is_barcode(s: AdapterSet) -> bool:
    return s.name.startswith("Barcode ")


def barcode_direction(s: AdapterSet) -> str:
    # decided on the START sequence's name alone (adapters.py:35-39)
    return "reverse" if "_rev" in s.start[0] else "forward"


def phase_b_pair_key(s: AdapterSet):
    """Barcode sets whose end-window score passes share a kernel in the pruned phase B (Pipeline._phase_b_pruned_records):
    barcodes 2k-1 and 2k of one direction -- two 24-mers scanned over the same windows read them once and run the
    one-stream kernel.  The pairing is a constant of the panel, so the pair kernels are built ahead of time
    (porechop_amd/aot.py).  -> a hashable key, None for a set that is scanned alone."""
    import re
    m = re.match(r"Barcode (\d+) \((forward|reverse)\)$", s.name)
    if not m:
        return None
    return (m.group(2), (int(m.group(1)) - 1) // 2)


def barcode_name(s: AdapterSet) -> str:
    """Shortest of the set name and its sequence names (first wins ties), spaces -> '_'."""
    names = [s.name]
    if s.start is not None:
        names.append(s.start[0])
    if s.end is not None:
        names.append(s.end[0])
    return min(names, key=len).replace(" ", "_")       # min() keeps the first of equal lengths, like sorted()[0]


def _barcode_set(panel: List[AdapterSet], number: int, direction: str) -> AdapterSet:
    want = "Barcode %d (%s)" % (number, direction)
    return next(s for s in panel if s.name == want)


def full_native_barcode(panel, number) -> AdapterSet:
    b = _barcode_set(panel, number, "reverse")
    return AdapterSet("Native barcoding %d (full sequence)" % number,
                      ("NB%02d_start" % number, "AATGTACTTCGTTCAGTTACGTATTGCTAAGGTTAA" + b.start[1] + "CAGCACCT"),
                      ("NB%02d_end" % number, "AGGTGCTG" + b.end[1] + "TTAACCTTAGCAATACGTAACTGAACGAAGT"))


_RAPID_TAIL = "GTTTTCGCATTTATCGTGAAACGCTTTCGCGTTTTTCGTGCGCCGCTTCA"


def full_rapid_barcode_old(panel, number) -> AdapterSet:      # SQK-RBK001
    b = _barcode_set(panel, number, "forward")
    return AdapterSet("Rapid barcoding %d (full sequence, old)" % number,
                      ("RB%02d_full" % number, "AATGTACTTCGTTCAGTTACG" + "TATTGCT" + b.start[1] + _RAPID_TAIL), None)


def full_rapid_barcode_new(panel, number) -> AdapterSet:      # SQK-RBK004
    b = _barcode_set(panel, number, "forward")
    return AdapterSet("Rapid barcoding %d (full sequence, new)" % number,
                      ("RB%02d_full" % number, "AATGTACTTCGTTCAGTTACG" + "GCTTGGGTGTTTAACC" + b.start[1] + _RAPID_TAIL), None)


def fix_up_1d2(matching: List[AdapterSet], score) -> List[AdapterSet]:
    """Drop 'SQK-MAP006 Short' when both 1D^2 parts score at least as well (score: set -> best
    start-or-end identity)."""
    by_name = {s.name: s for s in matching}
    if all(n in by_name for n in ("1D^2 part 1", "1D^2 part 2", "SQK-MAP006 Short")):
        sqk = score(by_name["SQK-MAP006 Short"])
        if score(by_name["1D^2 part 1"]) >= sqk and score(by_name["1D^2 part 2"]) >= sqk:
            return [s for s in matching if s.name != "SQK-MAP006 Short"]
    return matching


class NoBarcodes(Exception):
    pass


def choose_barcoding_kit(matching: List[AdapterSet], best_start, best_end) -> str:
    """'forward' or 'reverse' from the matching barcode sets' best scores (best_start/best_end:
    set -> identity).  Raises NoBarcodes with the reference's message when undecidable."""
    f_or = r_or = f_and = r_and = 0.0
    for s in matching:
        low = s.name.lower()
        if "barcode" not in low:
            continue
        bs, be = best_start(s), best_end(s)
        if "(forward)" in low:
            f_or += max(bs, be); f_and += bs; f_and += be
        elif "(reverse)" in low:
            r_or += max(bs, be); r_and += bs; r_and += be
    if f_or == 0 and r_or == 0:
        raise NoBarcodes("Error: no barcodes were found, so Porechop cannot perform barcode demultiplexing")
    if f_or > r_or:
        return "forward"
    if r_or > f_or:
        return "reverse"
    if f_and > r_and:
        return "forward"
    if r_and > f_and:
        return "reverse"
    raise NoBarcodes("Error: Porechop could not determine barcode orientation")


def add_full_barcode_sets(panel: List[AdapterSet], matching: List[AdapterSet]) -> List[AdapterSet]:
    names = {s.name for s in matching}          # membership is tested against the list as it was on entry
    out = list(matching)
    for i in range(1, 97):
        if "SQK-NSK007" in names and "Barcode %d (reverse)" % i in names:
            out.append(full_native_barcode(panel, i))
        if "Rapid" in names and "Barcode %d (forward)" % i in names:
            if "RBK004_upstream" in names:
                out.append(full_rapid_barcode_new(panel, i))
            elif "SQK-NSK007" in names:
                out.append(full_rapid_barcode_old(panel, i))
    return out
