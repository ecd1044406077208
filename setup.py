"""Build hook of the HIP extension: ``libmsda.so`` is compiled with hipcc for gfx950 (no torch
headers, plain C ABI) before the Python package is assembled.

    python setup.py build_hip          # just the library, in-tree
    pip install .                      # build_py runs build_hip first

Replaces the CUDA extension build the reference's container prepared for
(Containerfile:6-9 of the reference: FORCE_CUDA, TORCH_CUDA_ARCH_LIST)."""

import os
import subprocess
import sys
from pathlib import Path

from setuptools import Command, setup
from setuptools.command.build_py import build_py

ROOT = Path(__file__).resolve().parent
HIP_ARCH = "gfx950"


def hip_command(out: Path) -> list[str]:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return [
        hipcc, "-O3", f"--offload-arch={HIP_ARCH}", "-std=c++17", "-shared", "-fPIC",
        "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-pass-failed",
        "-o", str(out), str(ROOT / "unipercept_amd" / "csrc" / "msda.hip"),
    ]


class build_hip(Command):
    description = "compile unipercept_amd/libmsda.so with hipcc for gfx950"
    user_options = [("force", "f", "rebuild even when the library is newer than its sources")]

    def initialize_options(self):
        self.force = False

    def finalize_options(self):
        pass

    def run(self):
        out = ROOT / "unipercept_amd" / "libmsda.so"
        deps = list((ROOT / "unipercept_amd" / "csrc").glob("*")) + [ROOT / "include" / "msda.h"]
        if not self.force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
            print(f"{out} is up to date")
            return
        cmd = hip_command(out)
        print(" ".join(cmd))
        subprocess.run(cmd, check=True)


class build_py_with_hip(build_py):
    def run(self):
        self.run_command("build_hip")
        super().run()


if __name__ == "__main__":
    sys.path.insert(0, str(ROOT))
    setup(cmdclass={"build_hip": build_hip, "build_py": build_py_with_hip})
