#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
LPMP_PLAN_TIMES=1 LPMP_ROT_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "lpmp:" | head -60
