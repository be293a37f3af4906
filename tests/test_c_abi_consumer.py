"""The C ABI used from a C++ host program with no Python or torch in the process (tests/c_abi/consumer.cpp): it must
compile against include/spx_hip.h, link against libspx_hip.so, and - on a GPU - reproduce plain-loop results."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "consumer.cpp")


def _build(tmp_path):
    from scaleprotoseg_amd import _lib
    from scaleprotoseg_amd.build import build

    if not os.path.exists(_lib.LIB_PATH):
        build()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "spx_consumer")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = [hipcc, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", libdir, "-lspx_hip",
           f"-Wl,-rpath,{libdir}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return exe


def test_consumer_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_consumer_runs(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c-abi consumer ok" in r.stdout
