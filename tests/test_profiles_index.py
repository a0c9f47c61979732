"""profiles/INDEX.md lists every file under profiles/ (what it shows, what produced it, who cites it) and nothing else:
evidence that cannot be found is not evidence."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_profile_file_is_indexed_and_every_index_line_names_a_file():
    text = open(os.path.join(ROOT, "profiles", "INDEX.md")).read()
    named = set()
    for line in text.splitlines():
        if line.startswith("| ") and not line.startswith("| file") and not line.startswith("|---"):
            named |= set(re.findall(r"[A-Za-z0-9_.\-]+\.(?:txt|json|csv|npz|patch|md)", line.split("|")[1]))
    present = {f for f in os.listdir(os.path.join(ROOT, "profiles")) if f != "INDEX.md"}
    assert present - named == set(), "not in profiles/INDEX.md: %s" % sorted(present - named)
    assert named - present == set(), "indexed but missing: %s" % sorted(named - present)


def test_committed_pmc_summaries_parse_and_a_stale_one_is_only_a_warning():
    """bench.py reports roofline.traffic from the committed rocprofv3 PMC summary only while that summary names the sha256 of
    the kernel source in the tree (tools/prof_bench.sh writes it).  A tower edit without a new PMC run makes the driver's line
    say `traffic: null` with the reason — bench.py degrades safely, so here that state is a WARNING (an edit of the kernel
    source, a comment included, must not turn the CPU suite red until someone has a GPU box); a summary that does name the
    current source must parse to a plausible number."""
    import sys
    import warnings
    sys.path.insert(0, ROOT)
    import bench
    for streams in (2, 1):
        traffic, src = bench.measured_traffic(streams)
        if traffic is None:
            warnings.warn("roofline.traffic will be null for --streams %d: %s (re-run tools/prof_bench.sh on a GPU box)"
                          % (streams, src))
            continue
        assert traffic > 1e7 and src == bench.PMC_SUMMARY[streams], (streams, src)


def test_a_pmc_summary_of_another_kernel_source_is_not_reported(tmp_path, monkeypatch):
    """roofline.traffic goes null — and says why — when the committed summary was measured with another tower source."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    good = open(os.path.join(ROOT, bench.PMC_SUMMARY[2])).read()
    stale = tmp_path / "stale.txt"
    stale.write_text(re.sub(r"sha256: [0-9a-f]{64}", "sha256: " + "0" * 64, good))
    unnamed = tmp_path / "unnamed.txt"
    unnamed.write_text(re.sub(r"kernel source sha256: [0-9a-f]{64}[^\n]*\n", "", good))
    for path in (stale, unnamed):
        monkeypatch.setitem(bench.PMC_SUMMARY, 2, str(path))
        traffic, why = bench.measured_traffic(2)
        assert traffic is None and "stale" in why
    monkeypatch.setitem(bench.PMC_SUMMARY, 2, str(tmp_path / "missing.txt"))
    assert bench.measured_traffic(2) == (None, None)
