#!/usr/bin/env python3
"""bench.py — message updates/sec of the LP_MP dual block-coordinate-ascent sweep on MI355X.

One "step" = one LP::ComputePass (forward + backward sweep, reference include/LP_MP.h:869-887) over the
workload of BASELINE.json configs[2]: 1024x1024 grid MRF, 32 labels, dense pairwise tables, anisotropic
weights.  One message update = one executed receive or send.  Inputs are generated
in HBM before the timed region.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 1024] [--labels 32]
                    [--order colour_major|row_major] [--no-cpu-baseline]
                    [--workload c3|c4|c5] [--partitioner auto|metis|builtin] [--partition-file F]

N > 1: one process per GPU (self-launched, or under torch.distributed.run).  Every wait of the start-up is bounded: the rendezvous
(--rendezvous-timeout), one all_reduce and one all_to_all_single with the real split sizes before the timed region
(--collective-timeout; on expiry or on a wrong answer the rank says what it waited for and exits with code 3 — no retry after a
GPU call, no re-exec), the launcher (--launch-timeout).  The line of an N-rank run says where the time went: per rank and as max / mean the
compute and exchange time per pass (events around every pack -> collective -> unpack span, in an untimed repetition of the
timed passes), bytes and number of exchanges, the slowest rank, the partitioner, the cut, and a scaling model with its
assumptions (`scaling_model`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_STREAM_GBS = 6290.0  # what a copy kernel streams from HBM on this chip (same guide; r01 copy probe agrees)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--labels", type=int, default=32)
    ap.add_argument("--pairwise", default="dense", choices=["dense", "potts"])
    ap.add_argument("--order", default="colour_major", choices=["colour_major", "row_major"])
    ap.add_argument("--mode", default="anisotropic")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"],
                    help="c3: the headline grid (BASELINE.json configs[2], --grid/--labels/--pairwise/--order); c4: the random "
                         "sparse graph of configs[3] (--c4-nodes/--c4-edges/--c4-labels), partitioned across the ranks, every "
                         "rank generating only its own part in HBM; c5: configs[4], a Potts grid + 100 k labeling-list factors of "
                         "mixed arity in one factor graph (--c5-*), the variables partitioned across the ranks, run in lock step")
    ap.add_argument("--c5-grid", type=int, default=512)
    ap.add_argument("--c5-labels", type=int, default=8)
    ap.add_argument("--c5-edge-vars", type=int, default=150_000)
    ap.add_argument("--c5-triplets", type=int, default=70_000)
    ap.add_argument("--c5-quads", type=int, default=30_000)
    ap.add_argument("--c5-window", type=int, default=64, help="a labeling-list factor's members are drawn from this many consecutive edge variables")
    ap.add_argument("--c5-order", default="index", choices=["index", "colour_major", "suggested"],
                    help="--workload c5: the edge variables as inserted (index: local triples chain them — thousands of dependent levels "
                         "per sweep, latency-bound on any number of GPUs); colour_major: the generator inserts them colour by colour "
                         "(ordering.colour_major_order_hyper); suggested: the model as inserted, run in the order the engine suggests for "
                         "it (lpmp_plan_suggest_order applied as a chain of relations: one level per colour)")
    ap.add_argument("--c5-small", action="store_true",
                    help="--workload c5 in miniature (64x64 grid, 2 000 edge variables, 900 + 400 factors): the state after warmup + steps "
                         "passes is checked against the oracle's (tests/golden/c5_small.npz) — `oracle_check` in the line")
    ap.add_argument("--partitioner", default="auto", choices=["auto", "metis", "builtin"],
                    help="c4 / c5 on several GPUs: METIS (pymetis or a loadable libmetis) when present, else the built-in reverse "
                         "Cuthill-McKee + balanced KL partitioner (multi_gpu.graph_partition); the line names what ran")
    ap.add_argument("--partition-file", default=None,
                    help="c4 / c5 on several GPUs: variable -> rank from a file instead (raw int64 / int32 `.bin`, `.npy`, or text as "
                         "gpmetis writes it; multi_gpu.load_partition_file); entries in the order of the variables as the run numbers "
                         "them (c4: after --c4-order; c5: variables in factor order, or one entry per factor)")
    ap.add_argument("--overlap-exchange", action="store_true",
                    help="lock-step schedule on several GPUs: post the collective of an exchange behind the cut-adjacent records of a level and "
                         "await it only before the first reader of what it ships, the interior records of the level in between "
                         "(lockstep.LockstepSchedule.program_overlapped; same results bit for bit).  Pays where the cut is thin (C5, grids); "
                         "on C4 (60 %% of the edges cut) almost every record is cut-adjacent")
    ap.add_argument("--rendezvous-timeout", type=float, default=300.0, help="several ranks: seconds to wait for all ranks at the store")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="several ranks: bound on the self test before the timed region (one all_reduce, one all_to_all_single with the real "
                         "split sizes) and the time-out handed to init_process_group; on expiry the rank exits with code 3")
    ap.add_argument("--assume-exchange-latency-us", type=float, default=30.0, help="scaling_model: assumed cost of one all-to-all-v over xGMI beyond its bytes")
    ap.add_argument("--assume-exchange-GBps", type=float, default=400.0, help="scaling_model: assumed all-to-all bandwidth per rank over xGMI (7 links x 153 GB/s peak)")
    ap.add_argument("--c4-nodes", type=int, default=2_000_000)
    ap.add_argument("--c4-edges", type=int, default=10_000_000)
    ap.add_argument("--c4-labels", type=int, default=16)
    ap.add_argument("--c4-order", default="colour_major", choices=["colour_major", "index"],
                    help="--workload c4: variable order of the random graph, the same on any number of GPUs and under every schedule "
                         "(colour_major: ordering.colour_major_order, one dependent level per colour — 9 per directional sweep, 16 exchanges per "
                         "lock-step pass; index: the generator's own order — 33 levels, 60 exchanges)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", default="auto", choices=["auto", "overlap", "boundary", "lockstep"],
                    help="several GPUs: 'overlap' (grids in colour-major order; the default there) = every rank holds its strip plus "
                         "--ghost-rows rows of its neighbours and runs plain joined passes, one exchange per (ghost-rows / 2 - 1) passes: "
                         "the unpartitioned sweep bit for bit, gap 0 (overlap.py); 'lockstep' = the parts run the unpartitioned sweep level "
                         "by level with halo copies in between (lockstep.py; gap 0, one exchange per dependent level that reads across the "
                         "cut; any graph); 'boundary' = every part sweeps its own sub-problem, cut messages reconciled in a boundary step "
                         "(multi_gpu.py; a dual-bound gap of 0.1 - 2 %, few exchanges).  auto: overlap for colour-major grids, lockstep otherwise "
                         "(the exact schedules; C4 at full size in 8 parts: lock step 2.6 ms per pass and part, boundary steps 4.0)")
    ap.add_argument("--ghost-rows", type=int, default=12, help="overlap schedule: rows of each neighbour a rank holds (even; n passes between exchanges need 2 n + 2)")
    ap.add_argument("--rows-layout", default="auto", choices=["auto", "on", "off"],
                    help="dense pairwise factors as [table | m1 | m2] rows of an engine-private buffer (lpmp_set_rows_layout): one burst per "
                         "receive instead of a table and two single lines elsewhere.  auto: on for --workload c4 on one GPU")
    ap.add_argument("--compare-schedules", action="store_true",
                    help="several GPUs, c3 / c4: ALSO run the other multi-GPU schedules on the same model (same pass count, after the timed "
                         "region and every other leg) for the `schedules` entry of the line.  Off by default: it builds two more full-size "
                         "runners, and on a real N-GPU node nothing that is not the measurement should be able to cost the line")
    ap.add_argument("--no-compare-schedules", action="store_true", help="(accepted for older command lines: the comparison is off unless --compare-schedules)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path even at WORLD_SIZE 1: init_process_group(nccl = RCCL), the partitioned "
                         "sweep with its (empty) all_to_all_single exchanges, device all_reduce — what an N-GPU launch executes "
                         "first, runnable on a 1-GPU box (tests/test_bench_contract.py)")
    ap.add_argument("--cpu-sample-grid", type=int, default=512)
    ap.add_argument("--also-row-major", action="store_true", help="also time the row-major ordering (extra key)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="--gpus N without a launcher: print the rank processes that WOULD be started (command, environment) as one "
                         "JSON object and exit; nothing touches a GPU (tests/test_bench_contract.py)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="self-launched ranks: give up after this many seconds (0: never)")
    ap.add_argument("--prewarm-ms", type=float, default=0.0,
                    help="untimed passes (prewarm_ms / 8 of them) before the W warmup steps so that clocks and "
                         "power state have settled; 0 disables")
    return ap.parse_args()


def build_device_grid(torch, H, W, L, pairwise, order, seed, engine_mod, synthetic, stream_ptr):
    """Model structure on the host, costs generated directly in HBM (counter-based generator)."""
    import numpy as np
    dev = torch.device("cuda", torch.cuda.current_device())
    n = H * W
    n_e = len(synthetic.grid_edges(H, W)[0])
    if pairwise == "dense":
        m = synthetic.grid_model(H, W, L, order=order, seed=seed, device_const=True, compute_primal=True)
        const = torch.empty(n_e * L * L, dtype=torch.float64, device=dev)
        dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    else:
        m = synthetic.grid_model(H, W, L, pairwise="potts", order=order, seed=seed, compute_primal=True)
        const = torch.empty(n_e, dtype=torch.float64, device=dev)
        dual = torch.zeros(n * L + n_e * 2 * L, dtype=torch.float64, device=dev)
    engine_mod.synth_fill(const.data_ptr(), const.numel(), seed, n * L, stream_ptr)
    engine_mod.synth_fill(dual.data_ptr(), n * L, seed, 0, stream_ptr)
    torch.cuda.synchronize()
    return m, const, dual


def time_passes(torch, dist, eng, steps, warmup, world, prewarm_ms=0.0):
    if prewarm_ms > 0:
        # a fixed count (identical on every rank: the partitioned pass contains collectives); ~8 ms per pass on C3
        eng.compute_pass(max(2, int(prewarm_ms // 8)))
        torch.cuda.synchronize()
    if hasattr(eng, "prepare_passes"):               # what depends on the pass count of a call is built outside the timed region
        eng.prepare_passes(warmup); eng.prepare_passes(steps)
    eng.compute_pass(warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.compute_pass(steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def auto_schedule(args) -> str:
    """the default multi-GPU schedule: always an exact one.  overlap: 2-colour grids with an even number of rows per strip (plain
    joined passes over windows with ghost rows); anything else — grids in another order, the C4 graph — runs in lock step"""
    return "overlap" if args.workload == "c3" and args.order == "colour_major" and args.grid % 2 == 0 else "lockstep"


def close_runner(r):
    """engine and, for the lock-step drivers, the exchange plans (device arrays behind lpmp_halo_*) of a finished runner"""
    sw = getattr(r, "sweep", None)
    if sw is not None and hasattr(sw, "close"):
        sw.close()
    r.engine.close()


# ---- several ranks: bounded start-up, where the time went ------------------------------------------------------------------
class Watchdog:
    """`with Watchdog(seconds, what, rank):` — when the block is not left in time the process says which rank waited for what and
    EXITS with code 3 (os._exit from a timer thread: the main thread may sit inside a collective that never returns).  Never a
    retry, never a re-exec: the process has touched the GPU."""

    def __init__(self, seconds, what, rank):
        self.seconds, self.what, self.rank, self.timer = float(seconds), what, rank, None

    def _fire(self):
        sys.stderr.write(f"bench.py: rank {self.rank}: {self.what} did not finish within {self.seconds:.0f} s; exiting with code 3\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            import threading
            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def exit_startup(message):
    """a start-up step of an N-rank run failed or timed out (rendezvous, self test of the collectives): the message on stderr and
    exit code 3 — the same code the watchdogs use, so that a driver can tell "the ranks never got together" (3) from a crash (1)"""
    sys.stderr.write(message + "\n")
    sys.stderr.flush()
    sys.exit(3)


def rendezvous(args, torch, dist, identity, rank, world):
    """All ranks meet at the store of MASTER_ADDR : MASTER_PORT (torch's own env:// rendezvous — under torch.distributed.run that
    is the agent's store) within --rendezvous-timeout, tell each other which PHYSICAL device they sit on (``identity``:
    lpmp_device_identity, the PCI address — device ordinals say nothing when every rank has a visibility mask of its own), and
    only then choose the backend: "nccl" (= RCCL) when all ranks have a GPU of their own, gloo with CPU-staged exchanges when
    ranks share a device (RCCL refuses two ranks per device; smoke runs on the 1-GPU box — timings of such a run mean nothing).
    Returns (identities of all ranks, backend)."""
    import datetime
    timeout = datetime.timedelta(seconds=max(1.0, args.rendezvous_timeout))
    try:
        store, _, _ = next(dist.rendezvous("env://", rank, world, timeout=timeout))
        store.set(f"lpmp/device/{rank}", identity)
    except Exception as ex:
        exit_startup(f"bench.py: rank {rank}: no rendezvous at {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} within "
                     f"{args.rendezvous_timeout:.0f} s ({type(ex).__name__}: {ex})")
    idents = []
    for r in range(world):
        try:
            store.wait([f"lpmp/device/{r}"], timeout)
            idents.append(store.get(f"lpmp/device/{r}").decode())
        except Exception as ex:
            exit_startup(f"bench.py: rank {rank}: rank {r} did not show up at the rendezvous within {args.rendezvous_timeout:.0f} s ({type(ex).__name__})")
    backend = os.environ.get("LPMP_DIST_BACKEND") or ("nccl" if len(set(idents)) == world else "gloo")
    dist.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(10.0, args.collective_timeout)))
    return idents, backend


def collective_self_test(args, torch, dist, runner, rank, world):
    """before the timed region, under a wall-clock bound: one all_reduce and one all_to_all_single with the REAL split sizes of this
    runner's largest exchange (zeros: nothing is unpacked) — a communicator whose ranks cannot reach each other fails here, named"""
    t0 = time.perf_counter()
    with Watchdog(args.collective_timeout, "the self test of the collectives (all_reduce + all_to_all_single with the run's split sizes)", rank):
        dev = "cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
        t = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        if int(t.item()) != world:
            exit_startup(f"bench.py: rank {rank}: all_reduce of 1 over {world} ranks gave {t.item()}")
        counts = runner.exchange_counts() if hasattr(runner, "exchange_counts") else None
        shipped = None
        if counts is not None and getattr(runner, "comm", None) is not None:
            out_c, in_c = counts
            send = torch.zeros(int(sum(out_c)), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
            got = runner.comm.exchange(send, out_c, in_c)
            torch.cuda.synchronize()
            if got.shape[0] != int(sum(in_c)):
                exit_startup(f"bench.py: rank {rank}: all_to_all_single returned {got.shape[0]} doubles, expected {int(sum(in_c))}")
            shipped = int(sum(out_c)) * 8
    return {"all_reduce": "ok", "all_to_all_bytes_out": shipped, "seconds": time.perf_counter() - t0, "bound_s": args.collective_timeout}


def rank_stats_leg(torch, dist, runner, steps):
    """untimed repetition of the timed passes with every exchange bracketed by events (multi_gpu.ExchangeProbe), gathered from all
    ranks: {"per_rank": {key: [...]}, "max": {...}, "mean": {...}, "slowest_rank": r} with the keys compute_ms_per_pass,
    exchange_ms_per_pass, exchanges_per_pass, exchange_bytes_out_per_pass, exchange_bytes_in_per_pass, redundant_fraction,
    total_ms_per_pass"""
    from lp_mp_amd import multi_gpu as MG
    st = runner.probe_passes(steps)
    return MG.gather_rank_stats(dist, torch, getattr(runner, "comm", None), st)


def single_gpu_reference(args):
    """ms per pass of the SAME workload on one GPU from the latest committed bench line (profiles/): the t_1 of the scaling model.
    (workload, ms, file) or None"""
    import glob
    pats = {"c3": "r*bench_c3_default.json", "c4": "r*_c4_bench_c4_default.json", "c5": "r*_c5_bench_c5_*.json"}[args.workload]   # (c5: one line per edge-variable order)
    want = workload_name(args)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pats)), reverse=True):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        if d.get("n_gpus") == 1 and d.get("config", {}).get("workload") == want:
            return {"ms_per_pass": d["ms_per_step"], "source": os.path.relpath(f, ROOT),
                    "library_source_hash": (d.get("library") or {}).get("source_hash")}
    return None


def scaling_model(args, world, stats, measured_ms, shared_device):
    """What the first run on real hardware confirms or refutes, with its assumptions in the line:
        projected ms per pass = t_run + n_exchanges * latency + bytes / bandwidth
    t_run = the slowest rank's compute time per pass (measured here; on a shared device it is time-sliced and means nothing),
    n_exchanges and bytes = measured per pass, latency and bandwidth = --assume-exchange-latency-us / --assume-exchange-GBps.
    Strong scaling (c4, c5: the graph is fixed): projected_speedup = t_1 / projected; weak scaling (c3: one grid per GPU):
    projected_efficiency = t_1 / projected.  t_1 = this workload's single-GPU line in profiles/ (null when none is committed)."""
    mx = stats["max"]
    lam, bw = args.assume_exchange_latency_us * 1e-3, args.assume_exchange_GBps * 1e9
    nbytes = max(mx["exchange_bytes_out_per_pass"], mx["exchange_bytes_in_per_pass"])
    proj = mx["compute_ms_per_pass"] + mx["exchanges_per_pass"] * lam + nbytes / bw * 1e3
    t1 = single_gpu_reference(args)
    strong = args.workload != "c3"
    out = {"formula": "t_run + n_exchanges * latency + bytes / bandwidth", "t_run_ms": mx["compute_ms_per_pass"],
           "n_exchanges_per_pass": mx["exchanges_per_pass"], "exchange_bytes_per_pass_and_rank": nbytes,
           "assumed_latency_us_per_exchange": args.assume_exchange_latency_us, "assumed_GBps_per_rank": args.assume_exchange_GBps,
           "projected_ms_per_pass": proj, "measured_ms_per_pass": measured_ms, "measured_exchange_ms_per_pass": mx["exchange_ms_per_pass"],
           "t1_ms_per_pass": t1["ms_per_pass"] if t1 else None, "t1_source": t1["source"] if t1 else None,
           # t_1 comes from a committed line, i.e. from another box and possibly another build: the line says which
           "t1_library_source_hash": t1["library_source_hash"] if t1 else None,
           "t1_stale": None if not t1 else t1["library_source_hash"] != library_source_hash(),
           "kind": "strong" if strong else "weak",
           "note": "ranks share a device: t_run is time-sliced, the projection means nothing" if shared_device else None}
    if t1:
        key = "speedup" if strong else "efficiency"
        out["projected_" + key] = t1["ms_per_pass"] / proj
        out["measured_" + key] = t1["ms_per_pass"] / measured_ms
        if out["t1_stale"]:
            out["t1_note"] = (f"t_1 was measured with library {str(t1['library_source_hash'])[:12]}, the running one is {str(library_source_hash())[:12]}: "
                              f"{key} compares two builds — re-run `bench.py --workload {args.workload}` on one GPU and commit its line")
    return out


def workload_name(args):
    if args.workload == "c4":
        return (f"random sparse graph G({args.c4_nodes}, {args.c4_edges}), {args.c4_labels} labels, dense pairwise, {args.mode} weights, "
                f"{args.c4_order} variable order")
    if args.workload == "c5":
        g, L, ne, nt, nq, w = c5_shape(args)
        return (f"{g}x{g} Potts grid ({L} labels) + {ne} edge variables, {nt} triplet and {nq} quadruple labeling-list factors (window {w}), "
                f"one factor graph, {args.mode} weights, {args.c5_order} edge-variable order")
    return f"{args.grid}x{args.grid} grid per GPU, {args.labels} labels, {args.pairwise} pairwise, {args.mode} weights, {args.order} order"


def c5_shape(args):
    if args.c5_small:
        return 64, args.c5_labels, 2000, 900, 400, min(args.c5_window, 64)
    return args.c5_grid, args.c5_labels, args.c5_edge_vars, args.c5_triplets, args.c5_quads, args.c5_window


def c5_global_model(args, S):
    g, L, ne, nt, nq, w = c5_shape(args)
    m = S.c5_model(g, g, L, ne, nt, nq, seed=4, window=w, colour_edge_vars=args.c5_order == "colour_major")
    if args.c5_order == "suggested":
        # the model as inserted, in the order the ENGINE suggests for it through the C ABI (lpmp_plan_suggest_order on the host-only
        # plan), applied as a chain of relations through all factors — what a C++ caller does
        from lp_mp_amd.engine import Plan
        m = m.with_factor_order(Plan(m).suggest_order(4)[0])
    return m


def model_partition(args, torch, dist, MG, gm, world):
    """variable -> rank for a general model (c5): --partition-file, or the partitioner on rank 0, broadcast.  Returns
    (part per FACTOR as lockstep_model wants it, name of what made it)"""
    import numpy as np
    is_right = np.zeros(gm.n_factors, bool); is_right[gm.m_right] = True
    var = np.nonzero(~is_right)[0]
    if args.partition_file:
        try:
            part = MG.load_partition_file(args.partition_file, gm.n_factors, world)
        except ValueError:
            pv = MG.load_partition_file(args.partition_file, var.shape[0], world)
            part = np.zeros(gm.n_factors, np.int64); part[var] = pv
        return part, f"file {os.path.basename(args.partition_file)}"
    if world == 1:
        return np.zeros(gm.n_factors, np.int64), "none (1 part)"
    used = []
    def compute():
        p, how = MG.graph_partition_model(gm, world, method=args.partitioner, return_method=True)
        used.append(how)
        return p
    part = MG.broadcast_partition(torch, dist, gm.n_factors, None, compute)
    return part, MG.broadcast_string(dist, used[0] if used else None)


def library_source_hash():
    """hash of the sources, headers and flags the RUNNING liblpmp_engine.so was compiled from (lp_mp_amd/build.py writes it next
    to the library; build() rebuilds whenever it differs from the sources, so after build() it is also the hash of the sources)"""
    from lp_mp_amd import build as B
    try:
        return open(B.STAMP).read().strip()
    except OSError:
        return None


def pmc_traffic(kernel_name, args):
    """Bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    collected in separate runs of this same command, corrected as MI355X_MICROARCH.md prescribes; summarised
    by tools/pmc_traffic.py into profiles/).  These are the L2's FABRIC-side request counters: a read served by the
    256 MiB Infinity Cache counts like one served by HBM.
    Returns (bytes or None, source or the reason there is none).  A summary counts only if it was taken with THIS library: it
    carries the source hash of the library that ran under the profiler (tools/profile_collect.py), compared here with the running
    library's stamp — a kernel change without a re-profile gives `traffic: null`, not a stale ratio."""
    import glob
    if getattr(args, "workload", "c3") == "c5":
        return None, "no counter passes for C5 (latency-bound, a mix of kernel classes)"
    if getattr(args, "workload", "c3") == "c4":
        if not (args.c4_nodes == 2_000_000 and args.c4_edges == 10_000_000 and args.c4_labels == 16):
            return None, "counter passes exist for the full-size workload only"
        what = "c4_dense16"
    elif args.grid == 1024 and args.labels == 32 and args.pairwise == "dense" and args.order == "colour_major":
        what = "c3_dense32"
    else:
        return None, "counter passes exist for the full-size workload only"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{what}.json")))
    if not files:
        return None, "no counter summary committed"
    mine = library_source_hash()
    stale = None
    for f in reversed(files):                         # the latest summary taken on this kernel (C4: in this variable order)
        d = json.load(open(f))
        if what == "c4_dense16" and d.get("variable_order", "index") != getattr(args, "c4_order", "index"):
            continue
        if d.get("kernel") not in kernel_name:
            continue
        if d.get("library_source_hash") != mine:
            if stale is None:
                stale = (f"stale: {os.path.relpath(f, ROOT)} was taken with library {str(d.get('library_source_hash'))[:12]}, "
                         f"the running one is {str(mine)[:12]} — re-profile (tools/profile_round.sh)")
            continue
        b = d.get("fabric_bytes_per_launch_avg", d.get("hbm_bytes_per_launch_avg"))   # r01/r02 summaries used the second name
        if d.get("passes_per_launch"):                   # one chain launch = all passes of the call
            b = b / d["passes_per_launch"] * args.steps
        elif d.get("launches_per_chain_launch"):         # a deep sweep as ONE persistent launch: the summary is per step inside it
            b = b * d["launches_per_chain_launch"]
        return b, os.path.relpath(f, ROOT)
    return None, stale or "no counter summary for this kernel"


def dual_bound_gap_c4(torch, dist, args, mode, world, rank, schedule=None):
    """the same for the C4 workload: a 20 000-node / 100 000-edge graph of the same generator, partitioned by the same partitioner
    (--partitioner; a --partition-file lists the big graph's variables and cannot be applied to the miniature: gap_config says what
    partitioned it), same schedule, against its unpartitioned sweep on rank 0"""
    from lp_mp_amd import engine as E, multi_gpu as MG, synthetic as S
    n, m, L, passes = 20000, 100000, args.c4_labels, args.steps
    rank_of = None
    schedule = schedule or args.schedule
    if schedule == "lockstep":
        from lp_mp_amd import lockstep as LS
        sw = LS.LockstepGraph(torch, dist, n, m, L, mode, seed=1, order=args.c4_order, partitioner=args.partitioner,
                              overlap_exchange=getattr(args, "overlap_exchange", False))
        sw.boundary_every, sw.global_cut_fraction = "level that reads across the cut (lock step)", sw.cut_fraction
        rank_of = sw.rank_of
    else:
        sw = MG.GraphSweep(torch, dist, n, m, L, mode, seed=1, order=args.c4_order, partitioner=args.partitioner)
        rank_of = sw.rank_of
    sw.compute_pass(passes)
    lb_part = sw.lower_bound()
    out = None
    if rank == 0:
        e = E.Engine(torch.cuda.current_device())
        e.upload(S.counter_graph_model(n, m, L, 1, rank=rank_of))
        e.set_reparametrization(mode)
        e.compute_pass(passes)
        lb_ref = e.lower_bound()
        e.close()
        how = f"partitioner {sw.partitioner}" + (f" (the timed run took its partition from {os.path.basename(args.partition_file)}, which lists the big "
                                                 f"graph's variables)" if getattr(args, "partition_file", None) else "")
        out = {"dual_bound_gap": (lb_ref - lb_part) / abs(lb_ref), "gap_config": f"G({n}, {m}), {L} labels in {world} parts ({how}), {passes} passes, "
               f"boundary step every {sw.boundary_every}", "cut_fraction": sw.global_cut_fraction, "lb_partitioned": lb_part, "lb_unpartitioned": lb_ref}
    close_runner(sw)
    return out


def make_strip_runner(torch, dist, args, schedule, H, mode):
    """this rank's part of the (world * H) x H strip grid under one of the three multi-GPU schedules"""
    from lp_mp_amd import multi_gpu as MG
    if schedule == "overlap":
        from lp_mp_amd import overlap as OV
        g = max(4, min(args.ghost_rows, H - (H % 2)))
        return OV.OverlapStrips(torch, dist, H, H, args.labels, args.pairwise, mode, seed=1, g=g)
    if schedule == "lockstep":
        from lp_mp_amd import lockstep as LS
        return LS.LockstepStrips(torch, dist, H, H, args.labels, args.pairwise, args.order, mode, seed=1, overlap_exchange=getattr(args, "overlap_exchange", False))
    return MG.StripSweep(torch, dist, H, H, args.labels, args.pairwise, args.order, mode, seed=1, boundary_every="pass")


def describe_strip_runner(runner, schedule, world, H, W):
    if schedule == "overlap":
        return (f"{world} row strips of {H}x{W} with {runner.part.g} ghost rows per side (rank 0's window: rows {runner.window_rows[0]}..{runner.window_rows[1]}), "
                f"plain joined passes, one exchange per {runner.sweep.chunk} passes: the unpartitioned sweep of the {world * H}x{W} grid")
    if schedule == "lockstep":
        return f"{world} row strips of {H}x{W} in lock step (the unpartitioned sweep, {runner.halo_steps_per_pass():.1f} halo exchanges per pass)"
    return f"{world} row strips of {H}x{W}, cut-edge exchange once per pass"


def dual_bound_gap(torch, dist, args, mode, world, rank, schedule=None):
    """Second half of BASELINE.json's metric.  Same partition schedule, same RCCL exchange, on strips small enough
    that rank 0 can also run the UNPARTITIONED (world*g) x g grid: gap = (LB_unpartitioned - LB_partitioned) /
    |LB_unpartitioned| after the same number of passes."""
    from lp_mp_amd import engine as E, multi_gpu as MG, synthetic as S
    g, passes = 128, args.steps
    schedule = schedule or args.schedule
    sw = make_strip_runner(torch, dist, args, schedule, g, mode)
    sw.compute_pass(passes)
    lb_part = sw.lower_bound()
    out = None
    if rank == 0:
        if schedule == "overlap":                    # ONE global colour-major order: the single-GPU model of the whole grid
            m = S.grid_model(world * g, g, args.labels, pairwise=args.pairwise, order="colour_major", seed=1)
        else:                                        # strip-major order (the strips' own orders one after the other)
            ei, ej = MG.strip_global_edges(g, g, world, args.order)
            un, tables, potts = MG.strip_costs(g, g, args.labels, world, args.pairwise, 1)
            m = S.mrf_model(world * g * g, args.labels, ei, ej, un, tables=tables, potts=potts)
        e = E.Engine(torch.cuda.current_device())
        e.upload(m)
        e.set_reparametrization(mode)
        e.compute_pass(passes)
        lb_ref = e.lower_bound()
        e.close()
        out = {"dual_bound_gap": (lb_ref - lb_part) / abs(lb_ref), "gap_config": f"{world} strips of {g}x{g}, {passes} passes, schedule {schedule}",
               "lb_partitioned": lb_part, "lb_unpartitioned": lb_ref}
    close_runner(sw)
    return out


def dual_bound_gap_c5(torch, dist, args, mode, world, rank, schedule=None):
    """the same for the C5 workload: its miniature (--c5-small shape, same generator, same edge-variable order), partitioned like the
    big one and run in lock step, against the unpartitioned sweep on rank 0"""
    import types
    from lp_mp_amd import engine as E, multi_gpu as MG, lockstep as LS, synthetic as S
    small = types.SimpleNamespace(**dict(vars(args), c5_small=True, partition_file=None))
    gm = c5_global_model(small, S)
    part, how = model_partition(small, torch, dist, MG, gm, world)
    sw = LS.LockstepModel(torch, dist, gm, part, mode)
    sw.compute_pass(args.steps)
    lb_part = sw.lower_bound()
    out = None
    if rank == 0:
        e = E.Engine(torch.cuda.current_device())
        e.upload(gm)
        e.set_reparametrization(mode)
        e.compute_pass(args.steps)
        lb_ref = e.lower_bound()
        e.close()
        out = {"dual_bound_gap": (lb_ref - lb_part) / abs(lb_ref), "gap_config": f"{workload_name(small)} in {world} parts ({how}), {args.steps} passes, lock step",
               "cut_fraction": sw.cut_fraction, "lb_partitioned": lb_part, "lb_unpartitioned": lb_ref}
    close_runner(sw)
    return out


def golden_check_c5(torch, dist, args, runner, lb):
    """--c5-small: the state the timed passes left on the device(s) against the oracle's after the same number of passes
    (tests/golden/c5_small.npz, made by tests/golden/make_c5_small.py): the bound, and two exact checksums of the packed duals —
    on several ranks every rank adds up the factors it owns at their GLOBAL positions (integer sums wrap, so they add across ranks)"""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "c5_small.npz")
    if not (os.path.exists(path) and args.c5_labels == 8 and args.c5_window >= 64 and args.mode == "anisotropic" and args.prewarm_ms == 0):
        return None
    g = np.load(path)
    passes = args.warmup + args.steps
    hit = np.nonzero(g["passes"] == passes)[0]
    if hit.size == 0:
        return {"passes": passes, "note": "no oracle fixture for this pass count (fixture: 0..%d)" % int(g["passes"].max())}
    k, o = int(hit[0]), args.c5_order
    eng = getattr(runner, "engine", runner)
    eng.synchronize()
    d = eng.download_duals().view(np.uint64)
    part = getattr(runner, "part", None)
    with np.errstate(over="ignore"):
        if part is None:                                 # one engine holds the whole model
            w = np.arange(d.shape[0], dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            c = np.array([d.sum(dtype=np.uint64), (d * w).sum(dtype=np.uint64)], np.uint64)
        else:                                            # a lock-step part: owned factors at their global dual offsets
            from lp_mp_amd import synthetic as S
            import types
            gm = c5_global_model(types.SimpleNamespace(**vars(args)), S)
            goff, loff = gm.dual_offsets(), part.model.dual_offsets()
            own = np.nonzero(part.owned)[0]
            lens = (loff[own + 1] - loff[own]).astype(np.int64)
            first = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            within = np.arange(int(first[-1])) - np.repeat(first[:-1], lens)
            li = np.repeat(loff[own], lens) + within
            gi = (np.repeat(goff[part.factors_global[own]], lens) + within).astype(np.uint64)
            x = d[li]
            c = np.array([x.sum(dtype=np.uint64), (x * (gi * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64)], np.uint64)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.from_numpy(c.view(np.int64).copy())
        if dist.get_backend() != "gloo":
            t = t.cuda()
        dist.all_reduce(t)
        c = t.cpu().numpy().view(np.uint64)
    lbo = float(g[f"lb_{o}"][k])
    return {"passes": passes, "lb_oracle": lbo, "lb_rel_err": abs(lb - lbo) / max(1.0, abs(lbo)),
            "duals_bit_identical_to_oracle": (int(c[0]), int(c[1])) == (int(g[f"dual_sum_{o}"][k]), int(g[f"dual_wsum_{o}"][k])),
            "fixture": "tests/golden/c5_small.npz (oracle/lpmp_oracle.c on the same model, tests/golden/make_c5_small.py)"}


def hbm_min_bytes_per_pass(runner, args, world, bytes_per_pass, updates_per_pass, L):
    """Least HBM traffic of one pass: SURVEY 8(d) counts a dense table once per receive (8 L^2 per message and pass, two
    reads of every table per anisotropic pass); if every second read were served on-die, HBM would still deliver every
    table once + all vectors.  = algorithmic bytes - (table reads - tables) * 8 L^2.  Dense uniform-L workloads only."""
    if (args.workload == "c3" and args.pairwise != "dense") or args.workload == "c5":
        return None
    rows = args.grid * max(1, world)                 # N strips of grid x grid stacked: one (grid * N) x grid grid
    n_tables = args.c4_edges if args.workload == "c4" else rows * (args.grid - 1) + (rows - 1) * args.grid
    n_receives = updates_per_pass // 2              # a pass executes as many receives as sends
    return bytes_per_pass - (n_receives - n_tables) * 8 * L * L


def golden_check(torch, args, dual, lb):
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "c3_full_lb.npz")
    if not (args.grid == 1024 and args.labels == 32 and args.pairwise == "dense" and args.order == "colour_major"
            and args.mode == "anisotropic" and args.prewarm_ms == 0 and os.path.exists(path)):
        return None
    g = np.load(path)
    passes = args.warmup + args.steps
    hit = np.nonzero(g["passes_seed1"] == passes)[0]
    if hit.size == 0:                                # the fixture holds EVERY pass count 0..48 (make_c3_full.py)
        return {"passes": passes, "note": "no oracle fixture for this pass count (fixture: 0..%d)" % int(g["passes_seed1"].max())}
    k = int(hit[0])
    b = dual.view(torch.int64)
    w = torch.arange(b.numel(), dtype=torch.int64, device=b.device) * 2 + 1
    c0, c1 = int(b.sum().item()) & (2**64 - 1), int((b * w).sum().item()) & (2**64 - 1)
    lbo = float(g["lb_seed1"][k])
    return {"passes": passes, "lb_oracle": lbo, "lb_rel_err": abs(lb - lbo) / abs(lbo),
            "duals_bit_identical_to_oracle": (c0, c1) == (int(g["dual_sum_seed1"][k]), int(g["dual_wsum_seed1"][k])),
            "fixture": "tests/golden/c3_full_lb.npz (oracle/lpmp_oracle.c on the same inputs, tests/golden/make_c3_full.py)"}


def cpu_baseline(args, synthetic, M):
    """The oracle (single-threaded C restatement of the reference sweep; the reference sweep is
    single-threaded too) on a bounded sample of the same workload."""
    from oracle.binding import Oracle
    if getattr(args, "workload", "c3") == "c4":
        n, e = min(args.c4_nodes, 40000), min(args.c4_edges, 200000)
        m = synthetic.counter_graph_model(n, e, args.c4_labels, 1)
        what = f"G({n}, {e}), {args.c4_labels} labels, dense pairwise"
    elif getattr(args, "workload", "c3") == "c5":
        import types
        g, L, ne, nt, nq, w = c5_shape(args)
        a5 = types.SimpleNamespace(**dict(vars(args), c5_small=False, c5_grid=min(g, 128), c5_edge_vars=min(ne, 10000), c5_triplets=min(nt, 4600), c5_quads=min(nq, 2000)))
        m = c5_global_model(a5, synthetic)
        what = workload_name(a5)
    else:
        g = min(args.cpu_sample_grid, args.grid)
        m = synthetic.grid_model(g, g, args.labels, pairwise=args.pairwise, order=args.order, seed=1)
        what = f"{g}x{g} grid, {args.labels} labels, {args.pairwise} pairwise, {args.order}"
    o = Oracle(m)
    o.set_reparametrization(M.REPAM_NAMES[args.mode])
    o.ComputePass(1)
    r0, s0 = o.counters()
    passes = 0
    t0 = time.perf_counter()
    while True:
        o.ComputePass(1)
        passes += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or passes >= 50:
            break
    r1, s1 = o.counters()
    return {"value": (r1 - r0 + s1 - s0) / dt, "unit": "msg-updates/s", "cores": 1, "kind": "port",
            "sample": f"{what}, {passes} passes, oracle/lpmp_oracle.c single thread", "seconds": dt,
            "host_cpu": host_cpu_model(), "host_cores_available": os.cpu_count()}


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def rank_commands(args, argv, port):
    """the N rank processes of `python bench.py --gpus N` (no torchrun around it): same arguments, one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them"""
    cmds = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LPMP_BENCH_SELF_LAUNCHED": "1",
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        cmds.append({"argv": [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-run-launch"], "env": env})
    return cmds


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with WORLD_SIZE unset (the shape of the driver's single-GPU command with another N): this
    process becomes a launcher.  It never imports torch and never touches the GPU (a process that has initialised the GPU
    must not be replaced or forked into ranks): it starts N children, each of which is one rank on its own device, waits
    for all of them, relays rank 0's output (the JSON line last) and exits non-zero if any rank does."""
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmds = rank_commands(args, argv, port)
    if args.dry_run_launch:
        print(json.dumps({"launcher": "bench.py", "n_ranks": args.gpus, "ranks": cmds}), flush=True)
        return 0
    # the extension is compiled ONCE here (hipcc is a child process; no GPU call): the ranks then find a matching stamp
    from lp_mp_amd import build as B
    B.build()
    procs, outs = [], []
    tmp = tempfile.mkdtemp(prefix="lpmp_bench_")
    try:
        for r, c in enumerate(cmds):
            out = open(os.path.join(tmp, f"rank{r}.out"), "w+")
            outs.append(out)
            procs.append(subprocess.Popen(c["argv"], env=dict(os.environ, **c["env"]), stdout=out, stderr=None, cwd=ROOT))
        t0 = time.time()
        rc = 0
        live = set(range(len(procs)))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        print(f"bench.py launcher: rank {r} exited with code {code}", file=sys.stderr)
                        rc = code if code > 0 else 1
            if args.launch_timeout > 0 and time.time() - t0 > args.launch_timeout and live:
                print(f"bench.py launcher: ranks {sorted(live)} still running after {args.launch_timeout:.0f} s", file=sys.stderr)
                rc = 124
            time.sleep(0.05)
        for r in live:                                   # a rank failed: the others wait in a collective for ever — end them
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        for r, out in enumerate(outs):
            out.flush(); out.seek(0)
            text = out.read()
            if r == 0:
                sys.stdout.write(text)
            elif text.strip():
                sys.stderr.write(f"[rank {r} stdout]\n{text}")
        sys.stdout.flush()
        return rc
    finally:
        for out in outs:
            out.close()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run_launch):
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.schedule == "auto":
        args.schedule = auto_schedule(args)
    if args.schedule == "overlap" and (args.workload != "c3" or args.order != "colour_major"):
        print("bench.py: --schedule overlap is for grids in colour-major order (random graphs: lockstep or boundary)", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus != world:
        # the line would claim a GPU count that did not run (n_gpus comes from the process group)
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: start it as `python bench.py --gpus {args.gpus}` (it launches its own "
                  f"ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    from lp_mp_amd import build as B
    B.build_on_rank0(rank)                    # rebuilds when the sources changed; the other ranks wait for rank 0
    from lp_mp_amd import engine as E, model as M, synthetic as S

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    shared_device = False
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        n_dev = torch.cuda.device_count()                 # (counting devices does not initialise the GPU)
        if n_dev == 0:
            raise SystemExit("bench.py: no HIP device visible to rank %d (the engine has no CPU path)" % rank)
        torch.cuda.set_device(local_rank % n_dev)
        # which PHYSICAL device this rank sits on (PCI address + uuid through the C ABI): ordinals say nothing when the launcher
        # gives every rank a visibility mask of its own — then all of them report "device 0" and each has a GPU to itself
        identity = E.device_identity(torch.cuda.current_device())
        idents, backend = rendezvous(args, torch, dist, identity, rank, world)
        distinct = sorted(set(idents))
        shared_device = len(distinct) < world
        dev_coll = backend != "nccl" and bool(os.environ.get("LPMP_DIST_DEVICE_COLLECTIVES"))   # (multi_gpu.DistComm: no staging through the host)
        launch = {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend + (" (device tensors handed to the backend)" if dev_coll else ""),
                  "exchange_buffers": "device" if backend == "nccl" or dev_coll else "staged through the host", "ranks_seen": dist.get_world_size(),
                  "devices": [distinct.index(x) for x in idents], "device_identities": idents, "physical_devices": len(distinct),
                  "devices_visible": n_dev, "rendezvous_timeout_s": args.rendezvous_timeout,
                  "launcher": "bench.py (self-launched ranks)" if os.environ.get("LPMP_BENCH_SELF_LAUNCHED") else "external (torch.distributed.run)"}
        if launch["ranks_seen"] != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {launch['ranks_seen']} ranks")
        launch["persistent_launches"] = True
        if shared_device and not os.environ.get("LPMP_BENCH_KEEP_PERSISTENT"):   # (tools/shared_device_stall_probe.sh keeps them on)
            # ranks that SHARE a device (smoke runs on the 1-GPU box): the persistent chain launches assume that resident
            # workgroups keep running, which a device time-sliced between processes does not give them (kernels.hip, chain
            # executor: about every third 8-rank run stalled until its wait bound) — one launch per step instead.  The drivers
            # switch their own engine (lpmp_set_persistent_launches, DriverStats.own_the_engine finds the sharing itself); the
            # environment covers the auxiliary engines of the gap legs below
            os.environ["LPMP_NO_CHAIN"] = "1"
            os.environ["LPMP_NO_BLOCKED_PASSES"] = "1"
            launch["persistent_launches"] = False
            launch["note"] = "ranks share a device: persistent launches off, timings mean nothing"
    else:
        torch.cuda.set_device(0)
        launch = {"backend": None, "ranks_seen": 1, "devices": [0], "devices_visible": torch.cuda.device_count(), "launcher": "in-process (1 GPU)"}

    mode = M.REPAM_NAMES[args.mode]
    H = W = args.grid
    L = args.labels
    stream_ptr = torch.cuda.current_stream().cuda_stream

    dual = None
    setup = {}
    t_setup0 = time.perf_counter()
    if args.workload == "c4":
        from lp_mp_amd import multi_gpu as MG
        L = args.c4_labels
        part_of = MG.load_partition_file(args.partition_file, args.c4_nodes, world) if (args.partition_file and dist_on) else None
        if dist_on and args.schedule == "lockstep":
            from lp_mp_amd import lockstep as LS
            runner = LS.LockstepGraph(torch, dist, args.c4_nodes, args.c4_edges, L, mode, seed=1, order=args.c4_order, part_of=part_of,
                                      partitioner=args.partitioner, rows_layout=args.rows_layout == "on", overlap_exchange=args.overlap_exchange)
            parallelism = (f"{world} parts in lock step (the unpartitioned sweep in {args.c4_order} variable order, {runner.halo_steps_per_pass():.1f} halo exchanges per pass), "
                           f"{100 * runner.cut_fraction:.1f} % of the edges cut")
        else:
            rows = (not dist_on) and args.rows_layout != "off"
            runner = MG.GraphSweep(torch, dist if dist_on else None, args.c4_nodes, args.c4_edges, L, mode, seed=1, rows_layout=rows, order=args.c4_order,
                                   part_of=part_of, partitioner=args.partitioner)
            parallelism = (f"{world} parts, {100 * runner.global_cut_fraction:.1f} % of the edges cut, "
                           f"boundary step every {runner.boundary_every}") if world > 1 else "1 GPU"
        partitioner = (f"file {os.path.basename(args.partition_file)}" if part_of is not None else runner.partitioner) if dist_on else None
        updates_per_pass = runner.global_updates_per_pass
        bytes_per_pass = runner.global_bytes_per_pass
        levels = runner.levels
        eng = runner.engine
    elif args.workload == "c5":
        from lp_mp_amd import multi_gpu as MG, lockstep as LS
        if dist_on and args.schedule != "lockstep":
            raise SystemExit("bench.py: --workload c5 runs in lock step on several GPUs (--schedule lockstep)")
        L = args.c5_labels
        gm5 = c5_global_model(args, S)
        setup["model_on_host_s"] = time.perf_counter() - t_setup0
        if dist_on:
            part5, partitioner = model_partition(args, torch, dist, MG, gm5, world)
            runner = LS.LockstepModel(torch, dist, gm5, part5, mode, overlap_exchange=args.overlap_exchange)
            parallelism = (f"{world} parts in lock step (the unpartitioned sweep, {runner.halo_steps_per_pass():.1f} halo exchanges per pass), "
                           f"{100 * runner.cut_fraction:.1f} % of the message vectors cut")
            updates_per_pass = runner.global_updates_per_pass
            bytes_per_pass = runner.global_bytes_per_pass
            levels = runner.levels
            eng = runner.engine
        else:                                            # one GPU: the engine's own pass schedules ARE the sweep
            partitioner = None
            eng = E.Engine(torch.cuda.current_device())
            eng.set_stream(stream_ptr)
            eng.upload(gm5)
            eng.set_reparametrization(mode)
            eng.synchronize()
            runner = eng
            info = [eng.plan.schedule_info(d, mode) for d in (0, 1)]
            updates_per_pass = sum(i["n_receives"] + i["n_sends"] for i in info)
            bytes_per_pass = sum(i["algorithmic_bytes"] for i in info)
            levels = [i["n_levels"] for i in info]
            parallelism = "1 GPU"
    elif not dist_on:
        m, const, dual = build_device_grid(torch, H, W, L, args.pairwise, args.order, 1, E, S, stream_ptr)
        setup["model_structure_and_costs_in_hbm_s"] = time.perf_counter() - t_setup0
        t1 = time.perf_counter()
        eng = E.Engine(torch.cuda.current_device())
        eng.set_stream(stream_ptr)
        eng.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual), rows_layout=args.rows_layout == "on")
        eng.set_reparametrization(mode)
        eng.synchronize()
        setup["plan_schedules_upload_s"] = time.perf_counter() - t1   # ordering, weights, level schedule, op lists -> HBM
        runner = eng
        info = [eng.plan.schedule_info(d, mode) for d in (0, 1)]
        updates_per_pass = sum(i["n_receives"] + i["n_sends"] for i in info)
        bytes_per_pass = sum(i["algorithmic_bytes"] for i in info)
        levels = [i["n_levels"] for i in info]
        parallelism = "1 GPU"
    else:
        runner = make_strip_runner(torch, dist, args, args.schedule, H, mode)
        parallelism = describe_strip_runner(runner, args.schedule, world, H, W)
        updates_per_pass = runner.global_updates_per_pass
        bytes_per_pass = runner.global_bytes_per_pass
        levels = runner.levels
        eng = runner.engine
        partitioner = "row strips (closed form)"
    if not dist_on and args.workload == "c3":
        partitioner = None
    if dist_on:
        if hasattr(runner, "set_shared_device") and shared_device and not os.environ.get("LPMP_BENCH_KEEP_PERSISTENT"):
            runner.set_shared_device(True)
        launch["persistent_launches"] = bool(eng.persistent_launches)
        launch["self_test"] = collective_self_test(args, torch, dist, runner, rank, world)
    cut_fraction_line = float(getattr(runner, "global_cut_fraction", getattr(runner, "cut_fraction", 0.0))) if dist_on else None

    if getattr(runner, "setup_laps", None) is not None:
        setup.update(runner.setup_laps.laps)
    lb0 = runner.lower_bound()
    eng_rows = bool(getattr(eng, "rows_layout", False))
    setup["total_before_first_pass_s"] = time.perf_counter() - t_setup0
    t1 = time.perf_counter()
    if hasattr(runner, "prepare_passes"):            # ticket lists of the joined-pass chain launches (depends on the pass count)
        runner.prepare_passes(args.warmup); runner.prepare_passes(args.steps)
        setup["prepare_passes_s"] = time.perf_counter() - t1
    dt = time_passes(torch, dist, runner, args.steps, args.warmup, 2 if dist_on else 1, args.prewarm_ms)
    if updates_per_pass is None:                     # C4 on one GPU: what the line reports about the directional sweeps (multi_gpu.GraphSweep.query_info)
        t1 = time.perf_counter()
        runner.query_info()
        setup["directional_schedule_info_after_the_timed_region_s"] = time.perf_counter() - t1
        updates_per_pass, bytes_per_pass, levels = runner.global_updates_per_pass, runner.global_bytes_per_pass, runner.levels
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    lb1 = runner.lower_bound()
    # parity of exactly what was timed: the oracle ran ONCE on this model in the build container
    # (tests/golden/make_c3_full.py, seed 1); its lower bound and the checksums of its packed duals after the same
    # number of passes are compared with the state the timed passes left in HBM
    oracle_check = None
    if not dist_on and args.workload == "c3":
        eng.synchronize()                            # (rows layout: the packed dual buffer is written out here)
        oracle_check = golden_check(torch, args, dual, lb1)
    if args.workload == "c5" and args.c5_small:
        oracle_check = golden_check_c5(torch, dist if dist_on else None, args, runner, lb1)

    # roofline leg: the same passes again with every launch bracketed by HIP events on the engine's stream
    eng.reset_kernel_timing()
    eng.enable_kernel_timing(True)
    runner.compute_pass(args.steps)
    torch.cuda.synchronize()
    kt = eng.kernel_timing()
    eng.enable_kernel_timing(False)

    # several ranks: the same passes once more with every exchange bracketed by events — where the time of a pass goes, per rank
    rank_stats = model_line = None
    if dist_on:
        rank_stats = rank_stats_leg(torch, dist, runner, args.steps)
        model_line = scaling_model(args, world, rank_stats, dt / args.steps * 1e3, shared_device)

    # outside the timed region: one pass with primal rounding (what MpRoundingSolver runs every 5th iteration,
    # reference solver.hxx:387-397) and LP::EvaluatePrimal
    rounding = None
    if not dist_on and args.workload == "c3":
        # what a DEFAULT MpRoundingSolver run executes on every 5th iteration (standard_visitor.hxx:37,172-185): the visitor
        # switches to roundingReparametrization = damped_uniform (every message received and sent in EACH direction:
        # 2x the message updates of an anisotropic pass), then forward-and-primal, EvaluatePrimal,
        # backward-and-primal, EvaluatePrimal
        it = args.steps + args.warmup
        eng.set_reparametrization(M.REPAM_NAMES["damped_uniform"])
        eng.compute_pass_and_primal(it)                            # first call builds the schedules and the label-propagation lists
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.forward_pass_and_primal(it + 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cost_f = eng.evaluate_primal()
        t2 = time.perf_counter()
        eng.backward_pass_and_primal(it + 1)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        cost = eng.evaluate_primal()
        t4 = time.perf_counter()
        info_r = [eng.plan.schedule_info(d, M.REPAM_NAMES["damped_uniform"]) for d in (0, 1)]
        rounding = {"mode": "damped_uniform", "ms_forward_pass_and_primal": (t1 - t0) * 1e3, "ms_backward_pass_and_primal": (t3 - t2) * 1e3,
                    "ms_pass_and_primal": (t1 - t0 + t3 - t2) * 1e3, "ms_evaluate_primal": (t4 - t3) * 1e3,
                    "msg_updates_per_pass": sum(i["n_receives"] + i["n_sends"] for i in info_r),
                    "primal_cost_after_forward": cost_f, "primal_cost": cost, "lower_bound": eng.lower_bound()}
        eng.set_reparametrization(mode)
        # (the same under the weights of the timed passes, as rounds 1-3 reported it)
        eng.compute_pass_and_primal(it + 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.compute_pass_and_primal(it + 3)
        torch.cuda.synchronize()
        rounding["ms_pass_and_primal_" + args.mode] = (time.perf_counter() - t0) * 1e3

    gap_fn = {"c3": dual_bound_gap, "c4": dual_bound_gap_c4, "c5": dual_bound_gap_c5}[args.workload]
    gap = gap_fn(torch, dist, args, mode, world, rank) if dist_on else {"dual_bound_gap": 0.0, "gap_config": "1 GPU: the unpartitioned sweep itself"}
    # the other multi-GPU schedules on the same strips, same pass count (outside the timed region): what the choice costs
    schedules = None
    peak_bytes = torch.cuda.max_memory_allocated()
    if dist_on and args.workload == "c3" and args.compare_schedules and not args.no_compare_schedules:
        schedules = {args.schedule: {"ms_per_step": dt / args.steps * 1e3, "dual_bound_gap": gap["dual_bound_gap"] if gap else None, "timed": True}}
        # (the timed runner's 18.5 GB go first: N ranks may share one device in a smoke run)
        eng.close()
        for name in ("const", "dualt", "sweep", "engine"):
            if hasattr(runner, name):
                setattr(runner, name, None)
        eng = runner = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        for other in ("overlap", "lockstep", "boundary"):
            if other == args.schedule or (other == "overlap" and (args.order != "colour_major" or args.grid % 2)):
                continue
            r2 = make_strip_runner(torch, dist, args, other, H, mode)
            dt2 = time_passes(torch, dist, r2, args.steps, args.warmup, 2)
            t = torch.tensor([dt2], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lb2 = r2.lower_bound()
            close_runner(r2)
            del r2
            torch.cuda.empty_cache()
            g2 = dual_bound_gap(torch, dist, args, mode, world, rank, schedule=other)
            schedules[other] = {"ms_per_step": t.item() / args.steps * 1e3, "dual_bound_gap": g2["dual_bound_gap"] if g2 else None,
                                "lower_bound_after": lb2, "timed": False}
    if dist_on and args.workload == "c4" and args.compare_schedules and not args.no_compare_schedules:
        # C4: the exact schedule (lock step, colour-major variable order) beside the boundary-step one, same graph size
        from lp_mp_amd import multi_gpu as MG, lockstep as LS
        schedules = {args.schedule: {"ms_per_step": dt / args.steps * 1e3, "dual_bound_gap": gap["dual_bound_gap"] if gap else None, "timed": True}}
        eng.close()
        for name in ("const", "dualt", "sweep", "engine"):
            if hasattr(runner, name):
                setattr(runner, name, None)
        eng = runner = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        other = "boundary" if args.schedule == "lockstep" else "lockstep"
        # (the same partition as the timed run: same partitioner, same file)
        r2 = (MG.GraphSweep(torch, dist, args.c4_nodes, args.c4_edges, L, mode, seed=1, order=args.c4_order, part_of=part_of, partitioner=args.partitioner)
              if other == "boundary" else
              LS.LockstepGraph(torch, dist, args.c4_nodes, args.c4_edges, L, mode, seed=1, order=args.c4_order, part_of=part_of, partitioner=args.partitioner))
        dt2 = time_passes(torch, dist, r2, args.steps, args.warmup, 2)
        t = torch.tensor([dt2], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lb2 = r2.lower_bound()
        close_runner(r2)
        del r2
        torch.cuda.empty_cache()
        g2 = dual_bound_gap_c4(torch, dist, args, mode, world, rank, schedule=other)
        schedules[other] = {"ms_per_step": t.item() / args.steps * 1e3, "dual_bound_gap": g2["dual_bound_gap"] if g2 else None,
                            "lower_bound_after": lb2, "timed": False,
                            "note": "both schedules run the model in the same variable order (--c4-order); each gap is against the unpartitioned sweep of that model"}
    out = None
    if rank == 0:
        value = updates_per_pass * args.steps / dt
        dom = max(kt.items(), key=lambda kv: kv[1]["ms"]) if kt else None
        roof = None
        if dom:
            k = dom[1]
            avg_ms = k["ms"] / k["launches"]
            achieved = (k["bytes"] / k["launches"]) / (avg_ms * 1e-3) / 1e9
            traffic, src = pmc_traffic(k["kernel"], args) if not dist_on else (None, "one-GPU profiles only")
            hbm_min = hbm_min_bytes_per_pass(runner, args, world, bytes_per_pass, updates_per_pass, L)
            launch_s = avg_ms * 1e-3
            chain = bool(k.get("chain_launches"))
            roof = {"bound": "hbm", "kernel": k["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "avg_launch_ms": avg_ms,
                    "launches": k["launches"], "algorithmic_bytes_per_launch": k["bytes"] / k["launches"],
                    "passes_per_launch": args.steps / k["chain_launches"] if chain else None,
                    # `achieved` / `frac` are on the ALGORITHMIC scale (every table once per use, i.e. twice
                    # per pass).  The joined-pass chain launch serves the second read of a table from the 256 MiB
                    # Infinity Cache, so above HBM_STREAM_GBS the algorithmic rate is no longer an HBM rate and may even
                    # exceed the HBM peak: the honest HBM statement is the *_hbm_min_* triple below
                    "frac_scale": "algorithmic bytes (SURVEY 8d) / HBM spec peak",
                    "bound_note": (("algorithmic rate above what this chip streams from HBM (%.0f GB/s measured copy rate): the launch is "
                                    "bound by the L2 <-> Infinity-Fabric/Infinity-Cache path, not by HBM alone" % HBM_STREAM_GBS)
                                   if achieved > HBM_STREAM_GBS else "HBM-bound"),
                    "effective_bound": "infinity-cache/fabric" if achieved > HBM_STREAM_GBS else "hbm",
                    # what HBM must deliver at the very least: every pairwise table ONCE per pass + all message vectors / duals
                    "hbm_min_bytes_per_pass": hbm_min,
                    # the dominant kernel's launches at that minimum (same launch time): achieved * hbm_min / algorithmic
                    "hbm_min_GBps": None if hbm_min is None else achieved * hbm_min / bytes_per_pass,
                    # `traffic`: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch -- the L2's fabric-side request counters, which
                    # count Infinity-Cache hits like HBM reads (MI355X_MICROARCH.md); no HBM-side counter is in the recipe
                    "traffic_counts": "L2 fabric-side requests (Infinity-Cache hits included), not HBM-only bytes",
                    "fabric_traffic_frac_of_hbm_peak": None if traffic is None else traffic / launch_s / 1e9 / HBM_PEAK_GBS,
                    "traffic_over_algorithmic": None if traffic is None else traffic / (k["bytes"] / k["launches"])}
            if roof["hbm_min_GBps"] is not None:
                roof["hbm_min_frac"] = roof["hbm_min_GBps"] / HBM_PEAK_GBS
                roof["hbm_min_frac_of_measured_stream_rate"] = roof["hbm_min_GBps"] / HBM_STREAM_GBS
        out = {
            "metric": "message updates/sec + dual-bound gap, 32-label grid MRF @1/2/4/8 GPUs" if L == 32 and args.pairwise == "dense" and args.workload == "c3"
                      else "message updates/sec + dual-bound gap, " + {"c4": "random sparse graph MRF", "c5": "grid + labeling-list factors", "c3": "grid MRF"}[args.workload],
            "scaling_note": None if args.workload == "c3" else "strong scaling: the graph is fixed, the ranks share it",
            "value": value, "unit": "msg-updates/s", "n_gpus": world, "backend": launch["backend"], "ranks_seen": launch["ranks_seen"],
            "devices": launch["devices"], "launch": launch, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.workload == "c3" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "library": {"source_hash": library_source_hash(), "version": E.lib().lpmp_version().decode()},
            "config": {"workload": workload_name(args),
                       "parallelism": parallelism, "variable_order": {"c4": args.c4_order, "c5": args.c5_order, "c3": args.order}[args.workload],
                       "partitioner": partitioner, "cut_fraction": cut_fraction_line,
                       "pairwise_layout": "rows [table | m1 | m2], engine-private" if eng_rows else "packed (tables / serialize_dual order)",
                       "levels_per_direction": levels, "msg_updates_per_pass": updates_per_pass,
                       "algorithmic_bytes_per_pass": bytes_per_pass},
            "pass_algorithmic_GBps": bytes_per_pass * args.steps / dt / 1e9,
            "setup_s": setup, "peak_device_memory_GB_rank0": peak_bytes / 1e9,
            "lower_bound_before": lb0, "lower_bound_after": lb1,
            "oracle_check": oracle_check,
            "dual_bound_gap": gap["dual_bound_gap"], "dual_bound_gap_detail": gap, "schedule": args.schedule if dist_on else None,
            "overlap_exchange": bool(args.overlap_exchange) if dist_on and args.schedule == "lockstep" else None,
            "schedules": schedules,
            # several ranks: where a pass spends its time (untimed repetition under multi_gpu.ExchangeProbe), as max / mean over the
            # ranks and per rank; exchange spans include waiting for the slowest peer
            "compute_ms_per_pass": None if rank_stats is None else {"max": rank_stats["max"]["compute_ms_per_pass"], "mean": rank_stats["mean"]["compute_ms_per_pass"]},
            "exchange_ms_per_pass": None if rank_stats is None else {"max": rank_stats["max"]["exchange_ms_per_pass"], "mean": rank_stats["mean"]["exchange_ms_per_pass"]},
            # --overlap-exchange: the part of exchange_ms that is pack + copy + posting the collective (nothing hides it); 0 otherwise
            "exchange_post_ms_per_pass": None if rank_stats is None else {"max": rank_stats["max"].get("exchange_post_ms_per_pass"), "mean": rank_stats["mean"].get("exchange_post_ms_per_pass")},
            "exchange_bytes_per_pass": None if rank_stats is None else {"max": rank_stats["max"]["exchange_bytes_out_per_pass"], "mean": rank_stats["mean"]["exchange_bytes_out_per_pass"],
                                                                        "sum": sum(rank_stats["per_rank"]["exchange_bytes_out_per_pass"])},
            "exchanges_per_pass": None if rank_stats is None else rank_stats["max"]["exchanges_per_pass"],
            "redundant_fraction": None if rank_stats is None else rank_stats["max"]["redundant_fraction"],
            "slowest_rank": None if rank_stats is None else rank_stats.get("slowest_rank"),
            "rank_stats": rank_stats,
            "scaling_model": model_line,
            "kernels": kt,
            "rounding": rounding,
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:      # (the contract: the CPU baseline is timed on rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(args, S, M)
    if not dist_on and args.workload == "c3" and args.also_row_major and args.order != "row_major":
        del eng, runner
        m, const, dual = build_device_grid(torch, H, W, L, args.pairwise, "row_major", 1, E, S, stream_ptr)
        e2 = E.Engine(torch.cuda.current_device())
        e2.set_stream(stream_ptr)
        e2.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
        e2.set_reparametrization(mode)
        i2 = [e2.plan.schedule_info(d, mode) for d in (0, 1)]
        dt2 = time_passes(torch, dist, e2, args.steps, args.warmup, 1)
        upd2 = sum(i["n_receives"] + i["n_sends"] for i in i2)
        out["row_major"] = {"value": upd2 * args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3,
                            "levels_per_direction": [i["n_levels"] for i in i2], "lower_bound_after": e2.lower_bound()}
        e2.close()
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes its version banner through C stdio, which a pipe buffers
        # until exit — flush that first (seen on the GPU box: the banner landed after the line)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
