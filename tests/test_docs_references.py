"""The documents cite their evidence by file name: every profile, script, test file and source file named in DESIGN.md, README.md,
INTEGRATION.md and profiles/INDEX.md exists in the tree.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "INDEX.md"))


def _text():
    out = ""
    for d in DOCS:
        with open(os.path.join(ROOT, d)) as fh:
            out += fh.read()
    return out


def test_cited_profiles_exist():
    missing = set()
    for m in re.finditer(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_.\-]+\.(?:txt|csv|json|log))`", _text()):
        if not os.path.exists(os.path.join(ROOT, "profiles", m.group(1))):
            missing.add(m.group(1))
    assert not missing, sorted(missing)


def test_cited_scripts_tests_and_sources_exist():
    missing = set()
    # (examples/... and include/HPDDM_... are paths of the reference, cited as what ours replace: not checked here)
    for m in re.finditer(r"`((?:scripts|tests|oracle|hpddm_amd|include/hpddm_)[A-Za-z0-9_./\-]*\.(?:py|sh|hip|cpp|hpp|h|c))(?:::[A-Za-z0-9_\[\]\-]+)?`", _text()):
        if not os.path.exists(os.path.join(ROOT, m.group(1))):
            missing.add(m.group(1))
    assert not missing, sorted(missing)


def test_cited_test_functions_exist():
    missing = set()
    for m in re.finditer(r"`(tests/[A-Za-z0-9_]+\.py)::([A-Za-z0-9_]+)", _text()):
        path = os.path.join(ROOT, m.group(1))
        if not os.path.exists(path):
            missing.add(m.group(0))
            continue
        with open(path) as fh:
            if not re.search(r"def " + re.escape(m.group(2)), fh.read()):   # (a cited name may be the common prefix of a family: test_x_*)
                missing.add(m.group(0))
    assert not missing, sorted(missing)
