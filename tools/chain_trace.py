"""Latency budget of the chain executor from its per-ticket time stamps (engine.cpp, LPMP_CHAIN_TRACE).

    python tools/chain_trace.py run [grid] [labels] [order]   on the GPU box: one traced pass of a grid, then the analysis
                                                               (order c5: C5 with local triples instead of a grid)
    python tools/chain_trace.py show FILE                      analysis of a dump

Per ticket: t0 ticket in hand, t1 predecessors seen (wait over), t2 body done (stores issued), t3 published.
  startup  = t1 - t0 when the ticket did not have to wait (its predecessors were published before t0 + startup): the
             serial loads between knowing the ticket and being able to run it
  notice   = t1 - max(t3 of the predecessors) when it did wait: flag store -> poll sees it -> barrier
  body     = t2 - t1: dual loads, reduce, stores issued
  publish  = t3 - t2: stores acknowledged, barrier, flag store
The critical path of a level-by-level schedule is notice + body + publish per level."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    raw = open(path, "rb").read()
    n, nd = np.frombuffer(raw, np.int64, 2)
    o = 16
    st = np.frombuffer(raw, np.int64, 8 * n, o).reshape(n, 8); o += 64 * n
    tl = np.frombuffer(raw, np.int32, n, o); o += 4 * n
    off = np.frombuffer(raw, np.int32, n + 1, o); o += 4 * (n + 1)
    dep = np.frombuffer(raw, np.int32, nd, o)
    return st, tl, off, dep


def show(path):
    st, tl, off, dep = load(path)
    n = st.shape[0]
    us = lambda x: x * 0.01
    t0, t1, t2, t3, t4, t5 = st.T[:6]
    span = us(t3.max() - t0.min())
    n_launch = int(tl.max()) + 1
    print(f"{n} tickets in {n_launch} launches (levels), {span:.1f} us from first ticket to last publish: {span / n_launch:.3f} us per level")
    last_dep = np.full(n, -1, np.int64)
    has = off[1:] > off[:-1]
    # max publish time over the predecessors
    cnt = off[1:] - off[:-1]
    idx = np.repeat(np.arange(n), cnt)
    np.maximum.at(last_dep, idx, t3[dep])
    waited = has & (last_dep > t0)                   # the predecessor was published after the ticket was taken
    q = lambda a: "median %.2f, p10 %.2f, p90 %.2f us" % tuple(us(np.percentile(a, [50, 10, 90]))) if a.size else "-"
    print("tickets that had to wait:", int(waited.sum()), "of", n)
    print("  startup (no wait)   t1 - t0           :", q((t1 - t0)[~waited & has]))
    print("  lead    (waited)    last dep t3 - t0  :", q((last_dep - t0)[waited]))
    print("  notice  (waited)    t1 - last dep t3  :", q((t1 - last_dep)[waited]))
    print("  body                t2 - t1           :", q(t2 - t1))
    print("  publish             t3 - t2           :", q(t3 - t2))
    if t4.any():
        print("  body: duals landed  t4 - t1           :", q(t4 - t1))
        print("  body: receives      t5 - t4           :", q(t5 - t4))
        print("  body: sends         t2 - t5           :", q(t2 - t5))
    # per level: time from the level's first publish to the next level's first publish
    first_pub = np.full(n_launch, np.iinfo(np.int64).max, np.int64)
    np.minimum.at(first_pub, tl, t3)
    last_pub = np.zeros(n_launch, np.int64)
    np.maximum.at(last_pub, tl, t3)
    d = np.diff(last_pub)
    print("  level to level (last publish of consecutive launches):", q(d[d > 0]))
    # mailbox chains (plan.cpp): the hand-over between levels is a granule, not a flag — per level the medians of the stamps
    if t4.any() and n_launch > 8:
        med = lambda a: np.array([np.median(a[tl == l]) for l in range(n_launch)])
        m0, m1, m2, m4, m5 = med(t0), med(t1), med(t2), med(t4), med(t5)
        inner = slice(4, n_launch - 4)
        qq = lambda a: "median %.2f, p10 %.2f, p90 %.2f us" % tuple(us(np.percentile(a[inner], [50, 10, 90])))
        print("  per level (medians over the level's tickets):")
        print("    ready before the previous level's sends   t2[l-1] - t1[l] :", qq(m2[:-1] - m1[1:]))
        print("    hand-over   t4[l] - t2[l-1]  (sends issued -> vectors in) :", qq(m4[1:] - m2[:-1]))
        print("    receives    t5 - t4                                       :", qq(m5 - m4))
        print("    sends       t2 - t5                                       :", qq(m2 - m5))
        print("    level to level   t4[l] - t4[l-1]                          :", qq(np.diff(m4)))
        t6 = st.T[6]
        if t6.any():
            # a row-major grid, first half of a directional sweep: record i of anti-diagonal l reads what records i - 1 and i
            # of anti-diagonal l - 1 sent, so ticket b (4 records) waits for tickets b - 1 and b of the level before
            first = np.zeros(n_launch + 1, np.int64); np.add.at(first, tl + 1, 1); first = np.cumsum(first)
            hop, lead, grow = [], [], (first[1:] - first[:-1])
            for l in range(8, min(n_launch, 1000)):
                a0, a1, b0, b1 = first[l - 1], first[l], first[l], first[l + 1]
                if a1 - a0 < 2 or b1 - b0 < a1 - a0: continue
                nb = a1 - a0
                put = np.maximum(t6[a0:a1], np.concatenate([[0], t6[a0:a1 - 1]]))
                hop.append(t4[b0:b0 + nb] - put); lead.append(put - t1[b0:b0 + nb])
            hop = np.concatenate(hop); lead = np.concatenate(lead)
            print("    per ticket: sends issued (both producers) -> inputs landed   :", q(hop))
            print("    per ticket: waiting already when the producers' sends left   :", q(lead), " (negative: not ready yet: %.1f %%)" % (100.0 * np.mean(lead < 0)))
            print("    per ticket: landed -> own sends issued   t6 - t4              :", q(t6 - t4))
    wg = st.T[7]
    if wg.any():                                     # mailbox chains stamp the workgroup: how a workgroup spends the launch
        order = np.lexsort((t0, wg))
        same = wg[order][1:] == wg[order][:-1]
        prev, cur = order[:-1][same], order[1:][same]
        n_wg = len(np.unique(wg))
        total = float(t3.max() - t0.min()) * n_wg
        print(f"  {n_wg} workgroups; consecutive tickets of one workgroup are", "%d levels apart (median; p10 %d, p90 %d)" % tuple(np.percentile(tl[cur] - tl[prev], [50, 10, 90])))
        print("    share of a workgroup's time: start-up t1-t0 %.2f, waiting for inputs t4-t1 %.2f, arithmetic t2-t4 %.2f, publish t3-t2 %.2f, between tickets %.2f"
              % tuple(float(x.sum()) / total for x in (t1 - t0, t4 - t1, t2 - t4, t3 - t2, t0[cur] - t3[prev])))
        print("    ticket in hand -> published  t3 - t0 :", q(t3 - t0), "; published -> next ticket in hand :", q(t0[cur] - t3[prev]))
    # critical chain: follow the latest predecessor back from the last ticket
    k = int(np.argmax(t3)); hops = 0; parts = np.zeros(4)
    while True:
        b, e = off[k], off[k + 1]
        if e == b:
            break
        p = dep[b:e][np.argmax(t3[dep[b:e]])]
        if t3[p] > t0[k]:
            parts += [t1[k] - t3[p], t2[k] - t1[k], t3[k] - t2[k], 0]
        else:
            parts += [0, t2[k] - t1[k], t3[k] - t2[k], t1[k] - t0[k]]
        hops += 1; k = int(p)
    if hops:
        print(f"  critical chain back from the last ticket: {hops} hops; per hop notice {us(parts[0]) / hops:.2f}, body {us(parts[1]) / hops:.2f}, "
              f"publish {us(parts[2]) / hops:.2f}, startup not hidden {us(parts[3]) / hops:.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "show":
        show(sys.argv[2])
    else:
        g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
        order = sys.argv[4] if len(sys.argv) > 4 else "row_major"
        path = "/tmp/chain_trace.bin"
        code = f"""
import sys
sys.path.insert(0, {ROOT!r})
import torch
from lp_mp_amd import engine as E, model as M, synthetic as S
import bench as B
torch.cuda.set_device(0)
sp = torch.cuda.current_stream().cuda_stream
m, const, dual = B.build_device_grid(torch, {g}, {g}, {L}, "dense", {order!r}, 1, E, S, sp)
e = E.Engine(0); e.set_stream(sp)
e.upload(m, const_dev=const.data_ptr(), dual_dev=dual.data_ptr(), keep=(const, dual))
e.set_reparametrization(M.REPAM_ANISOTROPIC)
e.compute_pass(1); e.compute_pass(1); e.forward_pass()
e.synchronize()
"""
        env = dict(os.environ, LPMP_CHAIN_TRACE=path)
        if order == "c5":      # C5 with local triples: the ticket form of the generic chain kernel (LPMP_CHAIN_ALL=1), backward sweep
            env["LPMP_CHAIN_ALL"] = "1"
            code = f"""
import sys
sys.path.insert(0, {ROOT!r})
from lp_mp_amd import engine as E, model as M, synthetic as S
m = S.c5_model(512, 512, 8, 150000, 70000, 30000, seed=4, window=64)
e = E.Engine(0); e.upload(m); e.set_reparametrization(0)
e.compute_pass(1); e.backward_pass(); e.synchronize()
"""
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        show(path)
