"""bench.py on several single-GPU configurations, joined-pass chain on / off (LPMP_NO_BLOCKED_PASSES): which path each one
takes and what it costs.  python tools/config_ab.py ["--grid 1024 --labels 16" ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = sys.argv[1:] or ["--grid 1024 --labels 16", "--grid 1024 --labels 21", "--grid 2048 --labels 8", "--grid 1024 --labels 32 --mode uniform"]
for cfg in cfgs:
    for off in (False, True):
        env = dict(os.environ, LPMP_ROT_VERBOSE="1")
        if off:
            env["LPMP_NO_BLOCKED_PASSES"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "20", "--warmup", "3"] + cfg.split(),
                           env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        notes = sorted(set(l for l in r.stderr.splitlines() if l.startswith("lpmp:")))
        if not line:
            print(cfg, "FAILED", r.stderr[-300:]); continue
        d = json.loads(line[-1])
        print(f"{cfg:45s} {'launch per step' if off else 'default        '} {d['ms_per_step']:.4f} ms  {d['roofline']['kernel'][:44]}  {notes[-3:] if notes else ''}", flush=True)
