"""The R .Call shim cannot be built or run here (no R).  This only checks that it is valid C
against declarations of the R API entry points it uses and of our own header, so that a typo
cannot hide until someone builds it under R."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_r_shim_compiles_against_api_declarations(tmp_path):
    src = os.path.join(ROOT, "ldsr_amd", "r_shim", "ldsrhip_call.c")
    out = tmp_path / "shim.o"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "tests", "r_api_stub"),
                           "-I", os.path.join(ROOT, "include"), src])
    assert not out.exists()
