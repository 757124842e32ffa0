"""What a chain run that did not finish was doing (dumps of LPMP_CHAIN_TRACE, one per process): tickets in hand / past their
waits / past their bodies / published, and for the lowest unpublished tickets their state and that of their predecessors.
    python tools/chain_stall_report.py FILE..."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from chain_trace import load

for path in sys.argv[1:]:
    st, tl, off, dep = load(path)
    n = st.shape[0]
    t0, t1, t2, t3 = st.T[:4]
    print(f"{path}: {n} tickets; in hand {int((t0 != 0).sum())}, waits over {int((t1 != 0).sum())}, bodies done {int((t2 != 0).sum())}, published {int((t3 != 0).sum())}")
    open_ = np.nonzero((t0 != 0) & (t3 == 0))[0]
    if open_.size == 0:
        print("  no ticket is open"); continue
    print(f"  {open_.size} open tickets, lowest {open_[:8].tolist()}, highest drawn {int(np.nonzero(t0 != 0)[0].max())}")
    for t in open_[:6]:
        d = dep[off[t]:off[t + 1]]
        print(f"  ticket {t} (launch {tl[t]}): waits over {bool(t1[t])}, body done {bool(t2[t])}; deps {d.tolist()} published {[bool(t3[x]) for x in d]}"
              f"; held since {(st[:, :4].max() - t0[t]) * 0.01:.0f} us before the last stamp")
