#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out /tmp/ctrace
for attempt in 1 2 3 4 5 6; do
  rm -f /tmp/ctrace/*
  LPMP_ROT_EXPLICIT=1 LPMP_CHAIN_TIMEOUT_S=6 LPMP_CHAIN_TRACE=/tmp/ctrace/t_%p.bin timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --no-compare-schedules > gpurun_out/r4_8_b$attempt.json 2> gpurun_out/r4_8_b$attempt.err
  rc=$?
  echo "explicit+trace attempt $attempt rc=$rc"
  if [ $rc -ne 0 ]; then
    grep -h "EngineError" gpurun_out/r4_8_b$attempt.err | sort | uniq -c | head
    ls -la /tmp/ctrace/ | head -20
    for f in /tmp/ctrace/*.aborted; do python tools/chain_stall_report.py $f 2>&1 | head -30; done | tee gpurun_out/r4_8_stall_report.txt
    python - <<'PY' 2>&1 | tee -a gpurun_out/r4_8_stall_report.txt
import glob, re, sys
sys.path.insert(0, "tools")
from chain_trace import load
import numpy as np
errs = open(sorted(glob.glob("gpurun_out/r4_8_b*.err"))[-1]).read()
for m in re.finditer(r"ticket (\d+) waited for ticket (\d+): flag word (-?\d+), epoch (\d+), ([0-9.]+) s, (\d+) tickets drawn", errs):
    print("reported:", m.groups())
for f in glob.glob("/tmp/ctrace/*.aborted"):
    st, tl, off, dep = load(f)
    t0, t1, t2, t3 = st.T[:4]
    base = t0[t0 != 0].min()
    print(f, "tickets", st.shape[0], "span s", (st[:, :4].max() - base) * 1e-8)
    # tickets whose wait (t1 - t0) was long
    w = (t1 - t0) * 1e-8
    long_ = np.nonzero(w > 1.0)[0]
    print("  waits > 1 s:", long_.size, long_[:10].tolist())
    for t in long_[:5]:
        d = dep[off[t]:off[t + 1]]
        print(f"  ticket {t} launch {tl[t]}: in hand at {(t0[t]-base)*1e-8:.3f} s, wait over at {(t1[t]-base)*1e-8:.3f} s; deps {d.tolist()} published at {[round((t3[x]-base)*1e-8, 3) for x in d]} in hand at {[round((t0[x]-base)*1e-8,3) for x in d]} body done {[round((t2[x]-base)*1e-8,3) for x in d]}")
    # longest publish phases and bodies
    for name, a in (("body t2-t1", (t2 - t1) * 1e-8), ("publish t3-t2", (t3 - t2) * 1e-8), ("startup t1-t0", w)):
        k = int(np.argmax(a)); print(f"  longest {name}: {a[k]:.3f} s at ticket {k} (launch {tl[k]}), in hand at {(t0[k]-base)*1e-8:.3f} s")
PY
    break
  fi
done
