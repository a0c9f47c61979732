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
