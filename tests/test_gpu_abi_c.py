"""The drop-in boundary used from C: tests/c/abi_caller.c compiled with gcc against include/morna_hip.h alone and run as a
process of its own (no Python, no torch in it) -- AnnoyIndex-shaped entry points, exact search by vector and by item, the
row-sharded search on a communicator made through the C ABI, save / load, error codes.  -m gpu"""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_against_the_header_alone(tmp_path):
    exe = str(tmp_path / "abi_caller")
    libdir = os.path.join(ROOT, "morna_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_caller.c"), "-o", exe, "-L", libdir, "-lmorna_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    r = subprocess.run([exe, str(tmp_path / "x.annoy.mor")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi caller ok" in r.stdout
