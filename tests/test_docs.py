"""Documentation hygiene, checked instead of promised: DESIGN.md stays a design
document (<= 25 KiB; the lab notebook is EXPERIMENTS.md), every `profiles/...` file a document names
exists, and the measured tables of README.md / BASELINE.md are exactly what tools/results_table.py
writes from the committed bench lines and counter summaries -- a number in those tables cannot drift
from the evidence."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_is_a_design_document():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 25 * 1024
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert "@@" not in text                                   # no unfilled placeholder
    assert os.path.exists(os.path.join(ROOT, "EXPERIMENTS.md"))


def test_every_profiles_file_a_document_names_exists():
    missing = []
    for doc in ("DESIGN.md", "README.md", "BASELINE.md", "INTEGRATION.md", "EXPERIMENTS.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for m in set(re.findall(r"profiles/([A-Za-z0-9_.*{},-]+)", text)):
            name = m.rstrip(".,")
            if not name or name.endswith("_"):
                continue                                      # a prefix like profiles/r04_
            # brace and star patterns as the shell would expand them
            pats = [name]
            while any("{" in p for p in pats):
                nxt = []
                for p in pats:
                    b = re.search(r"\{([^{}]*)\}", p)
                    if not b:
                        nxt.append(p)
                        continue
                    for alt in b.group(1).split(","):
                        nxt.append(p[:b.start()] + alt + p[b.end():])
                pats = nxt
            for p in pats:
                p = re.sub(r"cfgN|_W_", lambda mm: "cfg*" if mm.group(0) == "cfgN" else "_*_", p)
                if not glob.glob(os.path.join(ROOT, "profiles", p)) and not glob.glob(os.path.join(ROOT, "profiles", p + "*")):
                    missing.append((doc, m))
    assert not missing, missing[:20]


def test_measured_tables_are_the_generated_ones():
    import results_table
    rounds = sorted({os.path.basename(f)[:3] for f in glob.glob(os.path.join(ROOT, "profiles", "r0*_cfg2_bench.json"))})
    block = results_table.table(rounds[-1])
    for name in ("README.md", "BASELINE.md"):
        s = open(os.path.join(ROOT, name)).read()
        a, b = "<!-- results:begin -->", "<!-- results:end -->"
        assert a in s and b in s, name
        assert s[s.index(a) + len(a):s.index(b)].strip() == block.strip(), "%s: run `python tools/results_table.py %s`" % (name, rounds[-1])
