"""Builds lp_mp_amd/csrc/liblpmp_engine.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO = os.path.join(CSRC, "liblpmp_engine.so")
SOURCES = ["kernels.hip", "engine.cpp", "plan.cpp"]
HEADERS = ["plan.hpp", os.path.join("..", "..", "include", "lpmp_engine.h"), os.path.join("..", "..", "include", "lpmp_model.h")]
# -ffp-contract=off: the sweep's duals must equal the sequential CPU semantics bit for bit
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall",
         "-Wno-unused-function"]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False) -> str:
    if force or needs_build():
        tmp = SO + ".tmp%d" % os.getpid()          # appear atomically: other ranks may be waiting for the file
        cmd = [hipcc()] + FLAGS + ["-o", tmp] + SOURCES
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, SO)
    return SO
