"""Rig-compiler policies that the C ABI does not report (CPU only): renderer_amd/csrc/rig_compile_selftest.cpp links the
rig compiler directly and looks at the CompiledRig — which morph pairs are stored dense, which compacted, and that a small
minority of compactable pairs is stored dense so the rig stays on the dense-only skin kernels."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "renderer_amd", "csrc")
EXE = os.path.join(ROOT, "renderer_amd", "lib", "rig_compile_selftest")


def test_morph_pair_layout_policy():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", CSRC, os.path.join("..", "lib", "rig_compile_selftest")], check=True, capture_output=True)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "rig compiler selftest ok" in p.stdout
