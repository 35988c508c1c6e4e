"""Build libapr_amd.so (HIP, gfx950) in-tree with hipcc.  No GPU needed to build."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libapr_amd.so")
SOURCES = ["engine.hip", "model_parse.cpp", "synth.cpp", "ingest.cpp"]
HEADERS = ["conv.hip.h", "frontend.hip.h", "wino.hip.h", "wino4.hip.h", "wino4_fused.hip.h", "wino_fused.hip.h", "wino_fused8.hip.h", "wino_s2.hip.h", "wino_s2_fused.hip.h", "report.hip.h", "resample.hip.h", "conv12_fused.hip.h", "model_parse.h", os.path.join("..", "..", "include", "apr.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(f) > t for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread",
           "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include"),
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
