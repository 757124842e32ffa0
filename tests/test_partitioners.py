"""Partitions of a multi-GPU run: the optional METIS binding (exercised against a TEST DOUBLE of libmetis — METIS itself is not in
this image), partitions handed in as files, and a hand-made partition through the lock-step driver in two gloo processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lp_mp_amd import model as M, multi_gpu as MG, synthetic as S  # noqa: E402
from oracle.binding import Oracle  # noqa: E402


@pytest.mark.parametrize("bits,layout", [(32, 51), (64, 51), (32, 52), (64, 52), (32, 0)])
def test_libmetis_binding_finds_the_index_width_and_the_option_layout(tmp_path, bits, layout):
    so = tmp_path / f"libfakemetis{bits}_{layout}.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", f"-DIDX_BITS={bits}", f"-DMETIS_LAYOUT={layout}",
                           os.path.join(ROOT, "tests", "cpp", "fake_metis.c"), "-o", str(so)])
    # (the library is looked up once per process: a child per width)
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "from lp_mp_amd import multi_gpu as MG, synthetic as S\n"
            "ei, ej = S.counter_graph_edges(3000, 9000, 1)\n"
            "p, how = MG.graph_partition(3000, ei, ej, 4, method='metis', return_method=True)\n"
            "q, how2 = MG.graph_partition(3000, ei, ej, 4, return_method=True)\n"
            "r, how3 = MG.graph_partition(3000, ei, ej, 4, seed=7, imbalance=0.1, return_method=True)\n"
            "print(json.dumps({'how': how, 'how_auto': how2, 'how_seeded': how3, 'sizes': np.bincount(p, minlength=4).tolist(), 'same': bool((p == q).all()),\n"
            "                  'contiguous': bool((np.diff(p) >= 0).all())}))\n" % ROOT)
    env = dict(os.environ, LPMP_METIS_LIB=str(so))
    env.pop("LPMP_PARTITIONER", None)
    d = json.loads(subprocess.check_output([sys.executable, "-c", code], env=env, text=True, timeout=300).strip().splitlines()[-1])
    assert d["how"].startswith("metis (") and f"{bits}-bit idx_t" in d["how"] and d["how_auto"] == d["how"] == d["how_seeded"]
    # the option slots are the library's own (metis.h 5.1 and 5.2 number them differently); a library that tells nothing gets none
    assert {51: "options in the 5.1 layout", 52: "options in the 5.2 layout", 0: "METIS' default options"}[layout] in d["how"]
    assert d["sizes"] == [750] * 4 and d["same"] and d["contiguous"]


def test_a_metis_that_fails_costs_auto_nothing_and_metis_an_error(tmp_path):
    so = tmp_path / "libbrokenmetis.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-DFAIL_ABOVE=64", os.path.join(ROOT, "tests", "cpp", "fake_metis.c"), "-o", str(so)])
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r)\n"
            "from lp_mp_amd import multi_gpu as MG, synthetic as S\n"
            "ei, ej = S.counter_graph_edges(3000, 9000, 1)\n"
            "p, how = MG.graph_partition(3000, ei, ej, 4, return_method=True)\n"
            "try:\n"
            "    MG.graph_partition(3000, ei, ej, 4, method='metis'); err = None\n"
            "except RuntimeError as ex:\n"
            "    err = str(ex)\n"
            "print(json.dumps({'how': how, 'sizes': np.bincount(p, minlength=4).tolist(), 'err': err}))\n" % ROOT)
    env = dict(os.environ, LPMP_METIS_LIB=str(so))
    env.pop("LPMP_PARTITIONER", None)
    d = json.loads(subprocess.check_output([sys.executable, "-c", code], env=env, text=True, timeout=300).strip().splitlines()[-1])
    assert d["how"].startswith("builtin") and "a METIS was found but failed" in d["how"] and "code -4" in d["how"]
    assert min(d["sizes"]) > 600 and "code -4" in d["err"]


def test_without_metis_auto_is_the_builtin_partitioner_and_metis_is_an_error():
    env_had = os.environ.pop("LPMP_METIS_LIB", None)
    try:
        MG._METIS.clear()
        ei, ej = S.counter_graph_edges(2000, 8000, 1)
        if MG.metis_partition(2000, ei, ej, 2) is not None:
            pytest.skip("a METIS is installed on this box")
        p, how = MG.graph_partition(2000, ei, ej, 4, return_method=True)
        assert how.startswith("builtin") and np.bincount(p, minlength=4).min() > 400
        with pytest.raises(RuntimeError, match="neither pymetis nor a loadable libmetis"):
            MG.graph_partition(2000, ei, ej, 4, method="metis")
        with pytest.raises(ValueError):
            MG.graph_partition(2000, ei, ej, 4, method="spectral")
    finally:
        if env_had is not None:
            os.environ["LPMP_METIS_LIB"] = env_had


def test_partition_files(tmp_path):
    part = (np.arange(50) * 7 % 3).astype(np.int64)
    for name, write in (("p.bin", lambda f: part.tofile(f)), ("p32.bin", lambda f: part.astype(np.int32).tofile(f)),
                        ("p.npy", lambda f: np.save(f, part)), ("p.txt", lambda f: np.savetxt(f, part, fmt="%d"))):
        f = str(tmp_path / name)
        write(f)
        assert np.array_equal(MG.load_partition_file(f, 50, 3), part), name
    f = str(tmp_path / "p.txt")
    with pytest.raises(ValueError, match="entries"):
        MG.load_partition_file(f, 51, 3)
    with pytest.raises(ValueError, match="parts must lie"):
        MG.load_partition_file(f, 50, 2)
    with pytest.raises(ValueError, match="without variables"):
        MG.load_partition_file(f, 50, 4)
    with pytest.raises(ValueError, match="neither"):
        MG.load_partition_file(str(tmp_path / "p.bin"), 49, 3)


WORKER = r"""
import os, sys, json, types, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import bench
from lp_mp_amd import model as M, multi_gpu as MG, lockstep as LS
from tests.mgpu_helpers import OracleEngine, materialise_fills
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, m, L = 300, 800, 3
part_of = MG.load_partition_file({part_file!r}, n, world)
def factory(part):
    materialise_fills(part)
    d = part.model.dual_data.copy()
    return torch.from_numpy(d), OracleEngine(part.model, d)
sw = LS.LockstepGraph(torch, dist, n, m, L, M.REPAM_ANISOTROPIC, seed=1, part_of=part_of, order="index", engine_factory=factory)
sw.compute_pass(3)
lb = sw.lower_bound()
stats = bench.rank_stats_leg(torch, dist, sw, 2)
args = types.SimpleNamespace(workload="c4", c4_nodes=n, c4_edges=m, c4_labels=L, c4_order="index", mode="anisotropic",
                             assume_exchange_latency_us=30.0, assume_exchange_GBps=400.0)
line = bench.scaling_model(args, world, stats, 1.25, False)
if rank == 0:
    json.dump({{"lb": lb, "stats": stats, "model": line, "partitioner": sw.partitioner, "cut": sw.cut_fraction,
               "exchange_counts": [c.tolist() for c in sw.exchange_counts()]}}, open(os.path.join({out!r}, "line.json"), "w"))
dist.destroy_process_group()
"""


def test_hand_made_partition_file_through_the_lock_step_driver_in_two_gloo_processes(tmp_path):
    """`bench.py --partition-file` in miniature on the CPU: a partition nobody's partitioner would produce (variable v on rank
    (7 v) mod 2) goes from a text file through load_partition_file into LockstepGraph (engine stand-in: the oracle), two gloo
    processes run the unpartitioned sweep with it — gap 0 — and the keys of the N-rank bench line come out per rank"""
    n, m, L = 300, 800, 3
    pf = tmp_path / "hand_made.part"
    np.savetxt(pf, (np.arange(n) * 7 % 2).astype(np.int64), fmt="%d")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path), part_file=str(pf)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)], env=env, cwd=ROOT, timeout=600)
    d = json.load(open(tmp_path / "line.json"))
    ref = Oracle(S.counter_graph_model(n, m, L, 1)); ref.set_reparametrization(M.REPAM_ANISOTROPIC); ref.ComputePass(3)
    assert abs(d["lb"] - ref.LowerBound()) <= 1e-12 * abs(ref.LowerBound())            # gap 0
    assert d["partitioner"] == "given" and 0.3 < d["cut"] < 0.7
    st = d["stats"]
    keys = {"compute_ms_per_pass", "exchange_ms_per_pass", "exchanges_per_pass", "exchange_bytes_out_per_pass", "exchange_bytes_in_per_pass",
            "redundant_fraction", "total_ms_per_pass", "exchange_post_ms_per_pass"}
    assert set(st["per_rank"]) == keys and all(len(v) == 2 for v in st["per_rank"].values())
    assert set(st["max"]) == keys and set(st["mean"]) == keys and st["slowest_rank"] in (0, 1)
    assert st["max"]["exchanges_per_pass"] >= 2 and st["max"]["exchange_bytes_out_per_pass"] > 0
    # what leaves rank 0 arrives at rank 1 and vice versa
    assert st["per_rank"]["exchange_bytes_out_per_pass"][0] == st["per_rank"]["exchange_bytes_in_per_pass"][1]
    for r in range(2):
        assert abs(st["per_rank"]["total_ms_per_pass"][r] - st["per_rank"]["compute_ms_per_pass"][r] - st["per_rank"]["exchange_ms_per_pass"][r]) < 1e-9
    mo = d["model"]
    assert mo["kind"] == "strong" and mo["t1_ms_per_pass"] is None and mo["measured_ms_per_pass"] == 1.25
    want = mo["t_run_ms"] + mo["n_exchanges_per_pass"] * 0.030 + mo["exchange_bytes_per_pass_and_rank"] / 400e9 * 1e3
    assert abs(mo["projected_ms_per_pass"] - want) < 1e-12
    out_c, in_c = d["exchange_counts"]
    assert out_c[0] == 0 and out_c[1] > 0 and in_c[1] > 0            # rank 0's largest exchange: everything goes to rank 1
