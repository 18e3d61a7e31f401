"""Register budget of the scan kernels that run BASELINE's T=1000 configs (compile-time check,
no GPU).  A refactor of the LDS accessors once cost the config-2 kernel 70 spilled VGPRs without
any test noticing, and the F1 -> F2 hand-over of e_t / B u_t spilled 24-73 VGPRs until F2 got a
scheduling barrier every 8 steps, so the budget is pinned: two waves per SIMD everywhere, no
scratch on the narrow kernels, and the wide config-3 kernel must not spill more than it does
today (23 VGPRs, 64 B).  Parses hipcc's
-Rpass-analysis=kernel-resource-usage through tools/resource_usage.py (~1 min)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_scan_kernel_register_budget():
    import resource_usage
    rows = {r[0]: r for r in resource_usage.table(os.path.join(ROOT, "ldsr_amd", "csrc", "em_scan_L16.hip"))}
    assert len(rows) == 48      # 16 padded (p, q) shapes x {static, queue, FIT}

    def get(name):
        (k,) = [k for k in rows if name in k]
        _, vgpr, agpr, vspill, scratch, occ, sgpr, sspill = rows[k]
        return vgpr, vspill, scratch, occ

    for tmpl in ("<1, 2, 16, 1, false, false, false>", "<1, 2, 16, 1, true, false, false>",
                 "<1, 4, 16, 1, false, false, false>", "<1, 1, 16, 1, false, false, false>",
                 "<2, 2, 16, 1, false, false, false>"):
        vgpr, vspill, scratch, occ = get(tmpl)
        assert (vspill, scratch, occ) == (0, 0, 2) and vgpr <= 256, (tmpl, vgpr, vspill, scratch, occ)
    vgpr, vspill, scratch, occ = get("<4, 8, 16, 1, false, false, false>")      # config 3
    assert occ == 2 and scratch <= 64, (vgpr, vspill, scratch, occ)
    for tmpl in ("<1, 2, 16, 1, false, false, true>", "<4, 8, 16, 1, false, false, true>"):   # FIT forms
        vgpr, vspill, scratch, occ = get(tmpl)
        assert (vspill, scratch, occ) == (0, 0, 2), (tmpl, vgpr, vspill, scratch, occ)


def test_pair_kernel_register_budget():
    """The two-cells-per-wave kernels of BASELINE configs 2 (T=1000: L=32) and 5 (T=813: L=26) and
    the four-cells-per-wave kernel of a short series, and the LEAD forms of configs 4 and 5: two waves per SIMD and no scratch (the first
    cut of the pair kernel spilled 73 VGPRs until the reverse composite moved into F2)."""
    import resource_usage
    for tu, tmpl in (("em_pair_L32.hip", "<1, 2, 32, 32, false, false>"), ("em_pair_L32.hip", "<1, 2, 32, 32, true, false>"),
                     ("em_pair_L26.hip", "<1, 4, 26, 32, false, false>"), ("em_quad_L13.hip", "<1, 2, 13, 16, false, false>"),
                     ("em_quad_L13.hip", "<1, 4, 13, 16, false, true>"), ("em_quad_L6.hip", "<1, 4, 6, 16, false, true>")):
        rows = resource_usage.table(os.path.join(ROOT, "ldsr_amd", "csrc", tu), tmpl)
        assert len(rows) == 1, (tu, tmpl, [r[0] for r in rows])
        _, vgpr, agpr, vspill, scratch, occ, sgpr, sspill = rows[0]
        assert (vspill, scratch) == (0, 0) and occ >= 2 and vgpr <= 256, (tu, tmpl, vgpr, vspill, scratch, occ)
