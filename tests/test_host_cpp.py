"""The C++ host side (renderer_amd/host): mirror of the reference's asset types, the scene
flattener that fixes the path's node ordering (DFS pre-order), and the RAII caller.  The checks
live in renderer_amd/host/selftest.cpp; this file runs that binary: host-only part on CPU, the
end-to-end part (through the C ABI, against a direct recursive evaluation) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "renderer_amd", "host")
EXE = os.path.join(HOST, "build", "host_selftest")


def selftest(*args):
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", HOST], check=True, capture_output=True)
    p = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host selftest ok" in p.stdout


def test_flattener_and_host_logic():
    selftest()


@pytest.mark.gpu
def test_scene_through_c_abi_matches_tree_evaluation():
    selftest("--gpu")
