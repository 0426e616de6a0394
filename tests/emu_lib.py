"""Test-only: the product's kernel source compiled for the CPU (tests/cpu_emu).  Debugging/sanitizer aid."""
import os
import subprocess

from raytracer_challenge_amd.backend import Backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "cpu_emu")
LIB = os.path.join(EMU_DIR, "_build", "librtc_emu.so")

_emu = None


def emu() -> Backend:
    global _emu
    if _emu is None:
        subprocess.run(["make", "-s", "-C", EMU_DIR], check=True)
        _emu = Backend(LIB)
        assert _emu.name == "hip-emu-cpu"
    return _emu
