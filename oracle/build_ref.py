"""oracle/build_ref.py — builds what of the reference itself compiles here, from the sources where they lie under /root/reference,
into oracle/_ref/ (git-ignored; travels to the GPU box like every built binary).  TEST INFRASTRUCTURE: nothing under lp_mp_amd/
uses it.

Buildable with plain g++ (no cmake, no external library, no generated code): `include/two_dimensional_variable_array.hxx` (the CSR
`weight_array` / `receive_array` container of the sweep) — the reference's own test of it
(`test/test_two_dimensional_variable_array.cpp` + `test/test.h`) and a driver of ours around the header (`oracle/ref_two_dim.cpp`);
`include/union_find.hxx` (numbers the partitions of the partition sweeps, a19) driven as `LP::construct_factor_partition` drives it
(`oracle/ref_union_find.cpp`); `two_smallest_elements` of `include/help_functions.hxx:106-120` (no third-party include), the scalar
two-minimum behind the Potts O(L) message, a13 (`oracle/ref_two_smallest.cpp`).
Everything else on the path (`LP_MP.h`, `factors_messages.hxx`, `vector.hxx`, `topological_sort.hxx` through `config.hxx`) needs
tclap / simdpp / meta from the empty `external/` submodules: unbuildable here, pinned by known answers instead."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("LPMP_REFERENCE", "/root/reference")
TARGETS = {"ref_two_dim": [os.path.join(HERE, "ref_two_dim.cpp")],
           "ref_union_find": [os.path.join(HERE, "ref_union_find.cpp")],
           "ref_two_smallest": [os.path.join(HERE, "ref_two_smallest.cpp")],
           "ref_test_two_dimensional_variable_array": [os.path.join(REF, "test", "test_two_dimensional_variable_array.cpp")]}


def available() -> bool:
    return os.path.exists(os.path.join(REF, "include", "two_dimensional_variable_array.hxx"))


def build(force: bool = False) -> dict:
    """-> {name: path} of the binaries that exist afterwards (built now if the reference is here, else whatever was built before)"""
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for name, src in TARGETS.items():
        exe = os.path.join(OUT, name)
        if available() and (force or not os.path.exists(exe) or any(os.path.getmtime(s) > os.path.getmtime(exe) for s in src)):
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "test")] + src + ["-o", exe])
        if os.path.exists(exe):
            out[name] = exe
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
