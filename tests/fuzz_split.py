"""Randomised parity with the ORACLE IN ANOTHER PROCESS, plus heap canaries in the engine's process.

Why: round 1 saw about one run in 50 000 where the oracle instance living in the same process as the HIP engine
produced different duals, later returned garbage and glibc aborted with `free(): invalid next size` — something wrote
into the host heap of the test process.  This harness separates the two suspects:

  * the oracle runs in a child process that never loads HIP (started before this process touches the GPU); models,
    calls and results travel over a pipe, so nothing the engine's process does can reach the oracle's memory;
  * the engine's process carries `--canaries` blocks of malloc'd memory of many sizes (the role the oracle's tables
    played as victims), filled with a pattern and verified after every seed, next to the guard regions the engine
    keeps around its own pinned buffers (engine.cpp, guarded_host_alloc) and glibc's MALLOC_CHECK_.

    python tests/fuzz_split.py FIRST COUNT [--minutes M] [--canaries N] [--families 0123]

Runs the test families of tests/test_fuzz_gpu.py (other seeds; --families 0123 by default, 4 = bipartite graphs with
joined multi-pass calls, 5 = pairwise factors that round themselves, 6 = labeling-list models in the level loop) until COUNT seeds or M minutes are over.
LPMP_STREAM_POOL=0 in the environment makes the engine create and destroy its HIP streams per engine (the round-1
behaviour) — the A/B switch of the hunt.  Exit code 0 = no event of any kind.
"""
import ctypes
import os
import pickle
import struct
import subprocess
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ---------------------------------------------------------------------------------------------------------------------
# child: oracle server (CPU only)
def _send(f, obj):
    b = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    f.write(struct.pack("<q", len(b))); f.write(b); f.flush()


def _recv(f):
    h = f.read(8)
    if len(h) < 8:
        raise EOFError
    (n,) = struct.unpack("<q", h)
    return pickle.loads(f.read(n))


def oracle_server():
    sys.path.insert(0, ROOT)
    from oracle.binding import Oracle
    inp, out = sys.stdin.buffer, os.fdopen(os.dup(1), "wb")
    os.dup2(2, 1)                                   # stray prints must not corrupt the reply stream
    objs = {}
    nxt = 0
    while True:
        try:
            req = _recv(inp)
        except EOFError:
            return
        try:
            if req[0] == "new":
                objs[nxt] = Oracle(req[1]); _send(out, ("ok", nxt)); nxt += 1
            elif req[0] == "del":
                objs.pop(req[1], None); _send(out, ("ok", None))
            elif req[0] == "call":
                _send(out, ("ok", getattr(objs[req[1]], req[2])(*req[3], **req[4])))
            elif req[0] == "modules":
                _send(out, ("ok", sorted(m for m in sys.modules if m.split(".")[0] in ("torch", "lp_mp_amd"))))
            else:
                _send(out, ("err", "unknown request"))
        except Exception as e:                      # noqa: BLE001 — reported to the caller, which re-raises
            _send(out, ("err", f"{type(e).__name__}: {e}"))


class RemoteOracle:
    """same methods as oracle.binding.Oracle, executed in the server process"""
    server = None

    def __init__(self, model):
        self._id = self._rpc(("new", model))

    @classmethod
    def _rpc(cls, req):
        _send(cls.server.stdin, req)
        st, val = _recv(cls.server.stdout)
        if st != "ok":
            raise RuntimeError(val)
        return val

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *a, **k: self._rpc(("call", self._id, name, a, k))

    def __del__(self):
        try:
            self._rpc(("del", self._id))
        except Exception:                           # noqa: BLE001 — interpreter shutdown
            pass


# ---------------------------------------------------------------------------------------------------------------------
# heap canaries in the engine's process
class Canaries:
    SIZES = (48, 200, 1000, 4096, 20000, 70000, 150000, 600000)

    def __init__(self, n):
        import numpy as np
        self.np = np
        self.libc = ctypes.CDLL("libc.so.6")
        self.libc.malloc.restype = ctypes.c_void_p
        self.libc.malloc.argtypes = [ctypes.c_size_t]
        self.libc.free.argtypes = [ctypes.c_void_p]
        self.blocks = []
        for i in range(n):
            self._add(i)
        self.bytes = sum(s for _, s, _ in self.blocks)

    def _add(self, i):
        size = self.SIZES[i % len(self.SIZES)]
        p = self.libc.malloc(size)
        tag = (i * 2654435761) & 0xFF
        ctypes.memset(p, tag, size)
        self.blocks.append((p, size, tag))

    def check(self):
        np = self.np
        bad = []
        for p, size, tag in self.blocks:
            a = np.frombuffer((ctypes.c_uint8 * size).from_address(p), np.uint8)
            if not (a == tag).all():
                idx = np.nonzero(a != tag)[0]
                bad.append((hex(p), size, int(idx[0]), int(idx[-1]), bytes(a[idx[0]:idx[0] + 16]).hex(), tag))
        return bad

    def churn(self, rng, frac=0.05):
        """free and re-allocate a few blocks so that the canaries stay interleaved with fresh heap activity"""
        for _ in range(max(1, int(len(self.blocks) * frac))):
            k = int(rng.integers(len(self.blocks)))
            p, size, tag = self.blocks[k]
            self.libc.free(p)
            q = self.libc.malloc(size)
            ctypes.memset(q, tag, size)
            self.blocks[k] = (q, size, tag)


class _DryRunEngine:
    """--dry-run: stands in for the HIP engine with an in-process oracle so that the harness itself (server, proxy,
    pickling, canaries) can be exercised on a machine without a GPU.  Never used for a parity claim."""

    def __init__(self, device=0):
        self.o = None
        self.rtype = 0

    def upload(self, m):
        from oracle.binding import Oracle
        self.m = m
        self.o = Oracle(m)
        self.o.set_reparametrization_type(self.rtype)

    def set_reparametrization_type(self, r):
        self.rtype = r
        if self.o is not None:
            self.o.set_reparametrization_type(r)

    def set_reparametrization(self, mode): self.o.set_reparametrization(mode)
    def compute_pass(self, n=1): self.o.ComputePass(n)
    def forward_pass(self): self.o.ComputeForwardPass()
    def backward_pass(self): self.o.ComputeBackwardPass()
    def lower_bound(self): return self.o.LowerBound()
    def download_duals(self): return self.o.duals()
    def compute_pass_custom(self, *rows): self.o.compute_pass_custom(*rows)
    def factor_lower_bounds(self):
        import numpy as np
        return np.array([self.o.factor_lower_bound(f) for f in range(self.m.n_factors)])

    def schedule_create(self, *rows, fuse=False):
        self.rows = rows
        return 0

    def schedule_run(self, sid): self.o.compute_pass_custom(*self.rows)
    def schedule_destroy(self, sid): pass
    def forward_pass_and_primal(self, it): self.o.ComputeForwardPassAndPrimal(it)
    def backward_pass_and_primal(self, it): self.o.ComputeBackwardPassAndPrimal(it)
    def compute_pass_and_primal(self, it): self.o.ComputePassAndPrimal(it)
    def download_primal(self): return self.o.primal()
    def check_primal_consistency(self): return self.o.CheckPrimalConsistency()
    def evaluate_primal(self): return self.o.EvaluatePrimal()
    def close(self): pass


class _Env:
    """stands in for pytest's monkeypatch in family 6 (the environment of this process is set for good)"""
    def setenv(self, k, v): os.environ[k] = v


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int)
    ap.add_argument("count", type=int)
    ap.add_argument("--minutes", type=float, default=1e9)
    ap.add_argument("--canaries", type=int, default=4000)
    ap.add_argument("--families", default="0123")
    ap.add_argument("--in-process-oracle", action="store_true", help="control run: the round-1 arrangement")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the harness with a stand-in engine")
    args = ap.parse_args()

    # the server starts BEFORE this process loads HIP: a child created later would be forked from a GPU process
    srv = None
    if not args.in_process_oracle:
        srv = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-server"], stdin=subprocess.PIPE,
                               stdout=subprocess.PIPE, env=dict(os.environ, MALLOC_CHECK_="3"))
        RemoteOracle.server = srv
        assert RemoteOracle._rpc(("modules",)) == [], "the oracle server must not load torch or the engine"

    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    import numpy as np
    import test_fuzz_gpu as T
    if srv is not None:
        T.Oracle = RemoteOracle
    if args.dry_run:
        T.E.Engine = _DryRunEngine
    fams = [T.test_random_models_all_modes_and_custom_passes, T.test_random_mrfs_fast_kernels_multi_pass_calls_and_fused_custom_schedules,
            T.test_random_mrfs_any_label_count_runtime_dims_kernels, T.test_random_mrfs_primal_rounding,
            T.test_random_bipartite_graphs_multi_pass_calls_equal_single_passes, T.test_random_mrfs_pairwise_factors_round_themselves,
            lambda seed: T.test_random_labeling_list_models_in_the_level_loop(seed, _Env())]
    fams = [fams[int(c)] for c in args.families]
    can = Canaries(args.canaries)
    rng = np.random.default_rng(args.first)
    t0 = time.time()
    events = {"mismatch": 0, "canary": 0, "engine_guard": 0, "other": 0}
    runs = 0
    print(f"fuzz_split: oracle {'in process' if srv is None else 'in server pid %d' % srv.pid}, {len(can.blocks)} canaries "
          f"({can.bytes / 1e6:.0f} MB), LPMP_STREAM_POOL={os.environ.get('LPMP_STREAM_POOL', '1')}, "
          f"MALLOC_CHECK_={os.environ.get('MALLOC_CHECK_', '')}", flush=True)
    seed = args.first
    while seed < args.first + args.count and time.time() - t0 < args.minutes * 60:
        for fn in fams:
            try:
                fn(seed)
            except AssertionError:
                events["mismatch"] += 1
                print("MISMATCH", fn.__name__, seed); traceback.print_exc(limit=2)
            except Exception as e:                  # noqa: BLE001
                key = "engine_guard" if "guard region" in str(e) else "other"
                events[key] += 1
                print("EVENT", key, fn.__name__, seed, e); traceback.print_exc(limit=2)
            runs += 1
        bad = can.check()
        if bad:
            events["canary"] += len(bad)
            print("CANARY OVERWRITTEN after seed", seed, bad[:8], flush=True)
            can = Canaries(args.canaries)
        can.churn(rng)
        seed += 1
        if (seed - args.first) % 200 == 0:
            print(f"  {seed - args.first} seeds, {runs} runs, {time.time() - t0:.0f} s, events {events}", flush=True)
    print(f"done: {seed - args.first} seeds, {runs} runs, {(time.time() - t0) / 60:.1f} minutes, events {events}", flush=True)
    if srv is not None:
        srv.stdin.close(); srv.wait(timeout=30)
        print("oracle server exit code", srv.returncode)
    sys.exit(1 if any(events.values()) else 0)


if __name__ == "__main__":
    if "--oracle-server" in sys.argv:
        oracle_server()
    else:
        main()
