#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for so in r03 pv1 pv2 pv3 default r03 default; do
  if [ $so = default ]; then unset LPMP_ENGINE_SO; else export LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_$so.so; fi
  python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
done
