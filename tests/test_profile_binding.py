"""bench.py reports counter traffic (roofline.traffic, traffic_ratio) only from a profile taken with the library it runs, or
with one built from the same device sources (tools/device_fingerprint.py): the fingerprint covers every source that decides
what the GPU does and nothing else."""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import device_fingerprint  # noqa: E402


def test_fingerprint_covers_kernels_and_launch_planning_but_not_host_io(tmp_path, monkeypatch):
    base = device_fingerprint.fingerprint()
    assert base == device_fingerprint.fingerprint() and len(base) == 40
    root = tmp_path / "repo"
    shutil.copytree(os.path.join(REPO, "porechop_amd", "csrc"), root / "porechop_amd" / "csrc",
                    ignore=shutil.ignore_patterns("*.o", "*.so", "*.tmp"))
    monkeypatch.setattr(device_fingerprint, "REPO", str(root))
    assert device_fingerprint.fingerprint() == base                      # object files and the like do not count
    for name, counts in (("pc_io.cpp", False), ("pc_gz.h", False), ("pc_kernels.hip", True), ("pc_api.cpp", True),
                         ("pc_jit_source.h", True), ("Makefile", True)):
        path = root / "porechop_amd" / "csrc" / name
        text = path.read_bytes()
        path.write_bytes(text + b"\n// touched\n" if name != "Makefile" else text + b"\n# touched\n")
        assert (device_fingerprint.fingerprint() != base) == counts, name
        path.write_bytes(text)
    assert device_fingerprint.fingerprint() == base


def test_recorded_summaries_name_what_they_were_taken_with():
    for name in sorted(os.listdir(os.path.join(REPO, "profiles"))):
        if name.endswith("_summary.json") and name >= "r05":
            sj = json.load(open(os.path.join(REPO, "profiles", name)))
            assert len(sj.get("library_sha1") or "") == 40, name
            assert len(sj.get("device_sources_sha1") or "") == 40, name
