#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
R03=$PWD/build/exp/liblpmp_engine_r03.so
for i in 1 2; do
  LPMP_ENGINE_SO=$R03 python tools/graph_time.py 200000 1000000 16 8 2>/dev/null | tail -1
  python tools/graph_time.py 200000 1000000 16 8 2>/dev/null | tail -1
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  python tools/row_major_time.py 1024 32 dense 10 2>/dev/null | tail -1
  LPMP_ENGINE_SO=$R03 python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
  python tools/row_major_time.py 512 8 potts 20 2>/dev/null | tail -1
done
