"""Register budget of the scan kernels that run BASELINE's T=1000 configs (compile-time check,
no GPU).  A refactor of the LDS accessors once cost the config-2 kernel 70 spilled VGPRs without
any test noticing, and the F1 -> F2 hand-over of e_t / B u_t spilled 24-73 VGPRs until F2 got a
scheduling barrier every 8 steps, so the budget is pinned: two waves per SIMD everywhere, no
scratch on the narrow kernels, and the wide config-3 kernel keeps ONE spilled 64-bit value (the
cell's result offset: stored before the EM loop, reloaded after it -- never inside; making the index
scalar or re-deriving it behind the loop removes the 12 bytes of scratch but costs SGPRs the sweeps
need: 170 -> 196 / 202 spilled, cfg3 +1.7 % on the same box, round 4).  Parses hipcc's
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
                 "<1, 4, 16, 1, false, false, false>", "<1, 1, 16, 1, false, false, false>"):
        vgpr, vspill, scratch, occ = get(tmpl)
        assert (vspill, scratch, occ) == (0, 0, 2) and vgpr <= 256, (tmpl, vgpr, vspill, scratch, occ)
    # (2,2) and (4,4): with the read-ahead ring of the image (SPF) a few values are spilled AROUND the EM loops
    # (test_scan_em_loops_have_no_scratch_traffic checks that they stay outside)
    for tmpl, ms, mb in (("<2, 2, 16, 1, false, false, false>", 4, 20), ("<4, 4, 16, 1, false, false, false>", 10, 28)):
        vgpr, vspill, scratch, occ = get(tmpl)
        assert occ == 2 and vspill <= ms and scratch <= mb, (tmpl, vgpr, vspill, scratch, occ)
    vgpr, vspill, scratch, occ = get("<4, 8, 16, 1, false, false, false>")      # config 3
    assert occ == 2 and vspill <= 4 and scratch <= 16, (vgpr, vspill, scratch, occ)
    for tmpl in ("<1, 2, 16, 1, false, false, true>", "<4, 8, 16, 1, false, false, true>"):   # FIT forms
        vgpr, vspill, scratch, occ = get(tmpl)
        assert (vspill, scratch, occ) == (0, 0, 2), (tmpl, vgpr, vspill, scratch, occ)


def test_scan_em_loops_have_no_scratch_traffic():
    """The EM loops (dense and masked body) of the short-chunk scan kernels with the read-ahead ring of the
    series image (SPF, em_scan_impl.h) hold no scratch instruction: what the allocator spills -- none to ten
    values depending on the shape -- is stored before a loop and reloaded behind it.  ((4,4) at L = 16, the
    widest shape with the ring at every chunk length, may keep ONE access in its masked loop; and the image reads
    do run ahead: no s_waitcnt lgkmcnt(0) directly behind a ds_read_b128 in F1 / B2 is what the ring is for,
    checked on the GPU by the small-launch timings of profiles/r04_small_launches.txt.)"""
    import subprocess
    import tempfile
    import loop_mix
    csrc = os.path.join(ROOT, "ldsr_amd", "csrc")
    shapes = [(1, 2, 16, "false", 0), (1, 2, 16, "true", 0), (2, 2, 16, "false", 0), (4, 4, 13, "true", 0),
              (4, 4, 16, "false", 1), (1, 4, 13, "true", 0)]
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "one.hip")
        with open(src, "w") as f:
            f.write('#include "em_scan_impl.h"\n')
            for pp, qq, L, q, _ in shapes:
                f.write("template __global__ void em_scan_kernel<%d, %d, %d, 1, %s, false, false>(EmParams);\n" % (pp, qq, L, q))
        asm = os.path.join(td, "one.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-I" + csrc, "--offload-arch=gfx950",
                        "--cuda-device-only", "-S", src, "-o", asm], check=True, capture_output=True)
        found = loop_mix.loops(asm, "em_scan_kernel")
    for pp, qq, L, q, allowed in shapes:
        tag = "ILi%dELi%dELi%dELi1ELb%dELb0ELb0EE" % (pp, qq, L, 1 if q == "true" else 0)
        big = [(hi - lo, c) for name, lo, hi, c, ops, n in found if tag in name and hi - lo >= 1000]
        assert big, tag
        assert max(c.get("scratch", 0) for _, c in big) <= allowed, (tag, [(n, c.get("scratch", 0)) for n, c in big])


def test_pair_kernel_register_budget():
    """The two-cells-per-wave kernels at the longest chunk without the steady-state form (L=23; from
    L=24 on the generic iterations of a fully observed series are a real function with its own stack: see
    test_steady_sweeps_have_no_spill_code) and
    the four-cells-per-wave kernel of a short series, and the LEAD forms of configs 4 and 5: two waves per SIMD and no scratch (the first
    cut of the pair kernel spilled 73 VGPRs until the reverse composite moved into F2)."""
    import resource_usage
    for tu, tmpl in (("em_pair_L23.hip", "<1, 2, 23, 32, false, false>"), ("em_pair_L23.hip", "<1, 2, 23, 32, true, false>"),
                     ("em_pair_L23.hip", "<1, 4, 23, 32, false, false>"), ("em_quad_L13.hip", "<1, 2, 13, 16, false, false>"),
                     ("em_quad_L13.hip", "<1, 4, 13, 16, false, true>"), ("em_quad_L6.hip", "<1, 4, 6, 16, false, true>")):
        rows = resource_usage.table(os.path.join(ROOT, "ldsr_amd", "csrc", tu), tmpl)
        assert len(rows) == 1, (tu, tmpl, [r[0] for r in rows])
        _, vgpr, agpr, vspill, scratch, occ, sgpr, sspill = rows[0]
        assert (vspill, scratch) == (0, 0) and occ >= 2 and vgpr <= 256, (tu, tmpl, vgpr, vspill, scratch, occ)


def test_steady_sweeps_have_no_spill_code():
    """BASELINE config 2's kernel (em_pair_kernel<1, 2, 32, 32>): the steady-state sweeps of fully
    observed series must run without scratch traffic.  The generic iterations of slow cells are a
    real function (pair_steady_g_phase, entered once per slow episode); every spill reload of the
    kernel has to sit in the basic block of that call, and the function's own iteration loop must be
    free of scratch traffic too.  (Both loops inlined in one body made the allocator spill ~150 VGPRs
    in whichever it took for the colder one: 38 k cycles per generic iteration against 22 k.)"""
    import re
    import subprocess
    import tempfile
    csrc = os.path.join(ROOT, "ldsr_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "one.hip")
        with open(src, "w") as f:
            f.write('#include "em_pair_impl.h"\n'
                    'template __global__ void em_pair_kernel<1, 2, 32, 32, false, false>(EmParams);\n'
                    'template __global__ void em_pair_kernel<1, 2, 32, 32, true, false>(EmParams);\n')
        asm = os.path.join(td, "one.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-I" + csrc, "--offload-arch=gfx950",
                        "--cuda-device-only", "-S", src, "-o", asm], check=True, capture_output=True)
        text = open(asm).read()
    kernels = re.split(r"\n(?=_Z14em_pair_kernel)", text)[1:]
    assert len(kernels) == 2
    # the G phase: one function per schedule; its EM loop (the blocks that do the sweeps' arithmetic) spill-free
    gfuncs = re.split(r"\n(?=_Z19pair_steady_g_phase)", text)[1:]
    assert len(gfuncs) == 2
    for g in gfuncs:
        g = g.split(".Lfunc_end")[0]
        for b in re.split(r"\n(?=\.LBB\d+_\d+:)", g):
            if len(re.findall(r"v_(?:fma|fmac|mul|add)_f64", b)) >= 20:
                assert "scratch_" not in b, b.split("\n")[0]
    for k in kernels:
        k = k.split(".Lfunc_end")[0]
        assert ".vgpr_count" not in k
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", k)
        calls = [b for b in blocks if "s_swappc_b64" in b]
        assert len(calls) == 1                                   # the fallback, nothing else
        for b in blocks:
            if "s_swappc_b64" in b:
                continue
            # every block that does the arithmetic of an EM iteration (the block that stores a
            # finished cell's results, once per cell, may reload what it stores)
            if len(re.findall(r"v_(?:fma|fmac|mul|add)_f64", b)) >= 20:
                assert "scratch_load" not in b, b.split("\n")[0]
    m = re.findall(r"\.vgpr_count:\s+(\d+)", text)
    assert m and all(int(x) <= 256 for x in m)


def test_config3_kernel_touches_no_scratch_inside_the_em_loop():
    """em_scan_kernel<4, 8, 16, 1> (BASELINE config 3): its 12 bytes of scratch hold one value
    that is stored before the EM loop and reloaded behind it; the loop bodies (dense and masked
    form) must not contain a scratch access, and no ds_read2 either (the series image is read with
    ds_read_b128 only since the odd value of a step is paired across two steps: the fused
    ds_read2st64_b64 of round 3 cost 8 LDS cycles plus 8 of bank conflicts per pair of steps)."""
    import re
    import subprocess
    import tempfile
    csrc = os.path.join(ROOT, "ldsr_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "one.hip")
        with open(src, "w") as f:
            f.write('#include "em_scan_impl.h"\n#include "ldsr_kernels.h"\n'
                    'template __global__ void em_scan_kernel<4, 8, 16, 1, false, false, false>(EmParams);\n')
        asm = os.path.join(td, "one.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-I" + csrc, "--offload-arch=gfx950",
                        "--cuda-device-only", "-S", src, "-o", asm], check=True, capture_output=True)
        lines = open(asm).read().split("\n")
    hdr = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
    loops = sorted(((hdr[i + 1] - hdr[i], hdr[i], hdr[i + 1]) for i in range(len(hdr) - 1)), reverse=True)[:2]
    assert len(loops) == 2 and loops[1][0] > 2000          # the two EM loops (masked and dense form)
    for _, a, b in loops:
        body = "\n".join(lines[a:b])
        # (the loop's exit path -- results of a finished cell -- follows the back edge in the listing)
        back = max(i for i in range(a, b) if re.search(r"s_c?branch\w* \.LBB0_%s$" % re.search(r"\.LBB0_(\d+):", lines[a]).group(1), lines[i]))
        body = "\n".join(lines[a:back])
        assert "scratch_" not in body
        assert "ds_read2" not in body and body.count("ds_read_b128") >= 200
