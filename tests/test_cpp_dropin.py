"""The C++ side of the drop-in boundary, compiled with g++ against the C ABI library:

  test_model_gpu     the standalone mirror (lp_mp_amd/include/LP_gpu.hxx + LP_gpu_solver.hxx) running the reference's
                     own test programs (test/test_model.cpp, test/graphical_model.cpp, test/multicut.cpp)
  test_offload_mock  lp_mp_amd/include/lpmp_offload.hxx taking a reference-SHAPED LP<FMC> (tests/cpp/mock_reference_lp.hxx:
                     the reference's member names and container surface, its own sweep absent) to the device through
                     serialize_dual, with kinds registered outside the ops; in the same translation unit as LP_gpu.hxx

Host-only parts run here, the full programs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROGRAMS = ["test_model_gpu", "test_offload_mock"]


def _build(tmp_path, name):
    from lp_mp_amd import build as B
    B.build()
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "lp_mp_amd", "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-L", B.CSRC, "-llpmp_engine", "-Wl,-rpath," + B.CSRC])
    return exe


@pytest.mark.parametrize("name", PROGRAMS)
def test_cpp_program_compiles_and_host_part_passes(tmp_path, name):
    exe = _build(tmp_path, name)
    out = subprocess.check_output([exe, "--host-only"], text=True)
    assert "all tests passed" in out


def test_offload_header_defines_no_reference_names():
    """lpmp_offload.hxx must be includable next to the reference's headers: it may not open namespace LP_MP"""
    import re
    src = open(os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_offload.hxx")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert not re.search(r"namespace\s+LP_MP\b", code)
    for hdr in ("LP_gpu.hxx", "LP_gpu_solver.hxx"):
        code = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "lp_mp_amd", "include", hdr)).read())
        opened = re.findall(r"namespace\s+(LP_MP\w*)\s*\{", code)
        assert opened and set(opened) <= {"LP_MP_gpu"}, opened      # LP_MP only ever appears as an alias, guarded


def test_offload_adapter_compiles_against_the_reference_headers(tmp_path):
    """Build container only: offloaded<LP_MP::LP<test_FMC>> + the reference's own Solver / StandardVisitor through
    g++ -fsyntax-only against /root/reference/include (tools/check_offload_against_reference.sh; a compile check, not an
    oracle).  The reference tree does not exist on the GPU box: skipped there."""
    if not os.path.isdir(os.environ.get("LPMP_REFERENCE", "/root/reference") + "/include"):
        pytest.skip("no reference tree on this box")
    log = str(tmp_path / "check.log")
    rc = subprocess.call([os.path.join(ROOT, "tools", "check_offload_against_reference.sh"), log])
    assert rc == 0, open(log).read()[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("name", PROGRAMS)
def test_cpp_program_full_run_on_device(tmp_path, name):
    exe = _build(tmp_path, name)
    out = subprocess.check_output([exe], text=True, timeout=600)
    assert "all tests passed" in out


def test_rccl_driver_builds_links_rccl_and_fails_loudly_without_a_gpu():
    """tools/mgpu_rccl_driver.cpp: the C++ host of the partitioned sweep (C ABI + <rccl/rccl.h>) compiles in build(), is
    linked against librccl and liblpmp_engine, and — like every product path — has no CPU fallback"""
    import torch
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("no <rccl/rccl.h> on this box")
    exe = B.build_mgpu_driver()
    src = open(os.path.join(ROOT, "lp_mp_amd", "include", "lpmp_multi_gpu.hxx")).read()
    assert "#include <rccl/rccl.h>" in src and "ncclGroupStart" in src and "ncclSend" in src and "ncclRecv" in src and "ncclAllReduce" in src
    needed = subprocess.check_output(["ldd", exe], text=True)
    assert "librccl" in needed and "liblpmp_engine" in needed
    if torch.cuda.is_available():
        pytest.skip("GPU box: the run itself is tests/test_multi_gpu.py::test_cpp_rccl_driver_equals_the_python_partitioned_sweep")
    r = subprocess.run([exe, "--H", "4", "--W", "4", "--L", "4"], capture_output=True, text=True, env=dict(os.environ, RANK="0", WORLD_SIZE="1"))
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_cpp_host_watchdog_exits_instead_of_waiting_for_ever(tmp_path):
    """lpmp_multi_gpu.hxx bounds ncclCommInitRank and the first all-reduce with exit_watchdog: a scope that is not left in time ends
    the process with code 3 and says which rank waited for what; a scope left in time costs nothing.  (Host-only program: nothing
    of RCCL or HIP is called.)"""
    from lp_mp_amd import build as B
    if not B.have_rccl():
        pytest.skip("<rccl/rccl.h> not installed")
    src = tmp_path / "wd.cpp"
    src.write_text('#include "lpmp_multi_gpu.hxx"\n#include <thread>\n'
                   'int main(int argc, char**) {\n'
                   '  { lpmp_mgpu::exit_watchdog ok(5.0, "rank 0 of 2: something quick"); }\n'
                   '  if (argc > 1) { lpmp_mgpu::exit_watchdog wd(0.3, "rank 1 of 2: ncclCommInitRank"); std::this_thread::sleep_for(std::chrono::seconds(20)); }\n'
                   '  std::puts("left in time"); return 0; }\n')
    exe = tmp_path / "wd"
    subprocess.check_call([B.hipcc(), "-std=c++17", "-O1", "-I", os.path.join(ROOT, "lp_mp_amd", "include"), str(src), "-o", str(exe), "-pthread"], timeout=600)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "left in time" in p.stdout
    p = subprocess.run([str(exe), "hang"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3 and "rank 1 of 2: ncclCommInitRank did not return within" in p.stderr
