#!/bin/bash
# tools/tile_sweep.sh — the joined-pass launch on shapes whose rows / slices are long: the engine's own choice (band or tiled ticket
# order) against forced tile sizes / depths and against one launch per step (tools/shape_probe.py; round 6)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
run() { # "shape args" label env...
  local shape=$1 label=$2; shift 2
  env "$@" LPMP_ROT_VERBOSE=1 timeout 900 python tools/shape_probe.py $shape 2> /tmp/tile_sweep.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$shape', '$label', round(d['ms_per_pass'],3), round(d['frac_of_8TBps'],3), list(d['kernels'].values()))"
  grep "tiles of about" /tmp/tile_sweep.err | tail -1 | cut -c1-200
  grep "passes as one launch\|stay one launch" /tmp/tile_sweep.err | tail -1 | cut -c1-200
}
for shape in "grid3d 96 32 16" "grid3d 128 16 16" "strip 256 4096 32 16" "strip 128 8192 32 16" "strip 3072 3072 32 16" "strip 2048 2048 32 16" "strip 1024 1024 32 20"; do
  run "$shape" "engine choice" A=1
  run "$shape" "launch by launch" LPMP_NO_BLOCKED_PASSES=1
  run "$shape" "bands only" LPMP_ROT_TILES=0
done
