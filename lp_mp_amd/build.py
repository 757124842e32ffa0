"""Builds lp_mp_amd/csrc/liblpmp_engine.so for gfx950 with hipcc (cross-compiles without a GPU).

``build()`` is cheap to call every time: the library carries a stamp (``liblpmp_engine.so.stamp``) with the hash of
the sources, headers and flags it was compiled from, and is rebuilt whenever that hash differs — so tests, bench and
smoke never run a binary that is older than the sources next to it (file times do not survive a copy to another box;
the stamp travels with the library)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import time

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO = os.path.join(CSRC, "liblpmp_engine.so")
STAMP = SO + ".stamp"
SOURCES = ["kernels.hip", "engine.cpp", "plan.cpp", "boundary.hip", "graph.cpp"]
HEADERS = ["plan.hpp", os.path.join("..", "..", "include", "lpmp_engine.h"), os.path.join("..", "..", "include", "lpmp_model.h")]
# -ffp-contract=off: the sweep's duals must equal the sequential CPU semantics bit for bit
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall",
         "-Wno-unused-function"]


def hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources() -> list:
    return [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]


def source_hash() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sources() + HEADERS:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not (os.path.exists(SO) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force: bool = False) -> str:
    if force or needs_build():
        want = source_hash()
        tmp = SO + ".tmp%d" % os.getpid()          # appear atomically: other ranks may be waiting for the file
        cmd = [hipcc()] + FLAGS + ["-o", tmp] + sources()
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, SO)
        with open(STAMP + ".tmp%d" % os.getpid(), "w") as fh:
            fh.write(want)
        os.replace(STAMP + ".tmp%d" % os.getpid(), STAMP)
    return SO


# ---- host programs on the C ABI (C++): the RCCL driver of the partitioned sweep ---------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MGPU_DRIVER = os.path.join(ROOT, "build", "mgpu_rccl_driver")
MGPU_SOURCES = [os.path.join(ROOT, "tools", "mgpu_rccl_driver.cpp"), os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_multi_gpu.hxx"),
                os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_overlap.hxx"), os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_lockstep.hxx"),
                os.path.join(ROOT, "include", "lpmp_engine.h"), os.path.join(ROOT, "include", "lpmp_model.h")]


def have_rccl() -> bool:
    """the RCCL development files the C++ multi-GPU host needs"""
    return any(os.path.exists(os.path.join(d, "rccl", "rccl.h")) for d in ("/opt/rocm/include", "/usr/include"))


def build_mgpu_driver(force: bool = False) -> str:
    """tools/mgpu_rccl_driver.cpp -> build/mgpu_rccl_driver (links liblpmp_engine.so and RCCL; hipcc only for its include
    and library paths — the file is plain host C++).  Stamped with a source hash like the library."""
    build()
    h = hashlib.sha256()
    for f in MGPU_SOURCES:
        with open(f, "rb") as fh:
            h.update(fh.read())
    want, stamp = h.hexdigest(), MGPU_DRIVER + ".stamp"
    if not force and os.path.exists(MGPU_DRIVER) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return MGPU_DRIVER
    os.makedirs(os.path.dirname(MGPU_DRIVER), exist_ok=True)
    tmp = MGPU_DRIVER + ".tmp%d" % os.getpid()
    subprocess.check_call([hipcc(), "-std=c++17", "-O2", "-Wall", MGPU_SOURCES[0], "-o", tmp, "-L", CSRC, "-llpmp_engine", "-lrccl",
                           "-Wl,-rpath," + CSRC])
    os.replace(tmp, MGPU_DRIVER)
    with open(stamp, "w") as fh:
        fh.write(want)
    return MGPU_DRIVER


def build_on_rank0(rank: int, timeout_s: float = 900.0) -> str:
    """multi-process launch (bench.py under torchrun): the first rank of every NODE compiles (LOCAL_RANK 0: nodes need not
    share a file system), the others wait for a matching stamp"""
    if int(os.environ.get("LOCAL_RANK", rank)) == 0:
        return build()
    t0 = time.time()
    while needs_build():
        if time.time() - t0 > timeout_s:
            raise RuntimeError("timed out waiting for rank 0 to build the HIP extension")
        time.sleep(0.5)
    return SO
