#!/bin/bash
# tools/sanitize_host.sh [out.txt] — sanitizer runs of the HOST side of the engine in the CPU container (GPU sanitizers are not available
# on this pool): the library built from the same sources with the sanitizer on the host compile (kernels unchanged), selected with
# LPMP_ENGINE_SO, the runtime preloaded into python.  ASan + UBSan over the host-logic test files (incl. the gloo worker processes),
# TSan with 8 planning threads over the same files without the torch.distributed workers (torch's own ProcessGroupGloo reports races
# on its condition variables under TSan), plus the C4-shaped graph through the C++ colouring / partition refinement / planner.
cd "$(dirname "$0")/.." || exit 1
OUT=${1:-profiles/r05_sanitizers_host.txt}
mkdir -p build/exp
SRC="kernels.hip engine.cpp plan.cpp boundary.hip graph.cpp"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-strict-aliasing -fno-omit-frame-pointer -Wno-unused-function"
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)")
FILES="tests/test_plan_host.py tests/test_graph_host.py tests/test_partitioners.py tests/test_lockstep.py tests/test_overlap.py tests/test_multi_gpu.py tests/test_lp_mirror.py tests/test_bench_contract.py tests/test_oracle_ref.py"
{
echo "# Sanitizer runs of the HOST side of the engine (round 6; CPU container, no GPU call), sources of $(git rev-parse --short HEAD) + working tree"
echo "## AddressSanitizer + UndefinedBehaviorSanitizer"
( cd lp_mp_amd/csrc && hipcc $FLAGS -fsanitize=address,undefined -o ../../build/exp/liblpmp_engine_asan.so $SRC ) 2>&1 | tail -3
LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0 LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_asan.so \
  timeout 3000 python -m pytest $FILES -q -m "not gpu" -p no:cacheprovider > build/exp/asan_pytest.log 2>&1
grep -v "Gloo\|^$" build/exp/asan_pytest.log | tail -4
echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' build/exp/asan_pytest.log)   UBSan reports: $(grep -c 'runtime error:' build/exp/asan_pytest.log)"
LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0 LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_asan.so python - <<'PY' 2>&1 | tail -5
import numpy as np
from lp_mp_amd import engine as E, synthetic as S, multi_gpu as MG, model as M
n, m = 200000, 1000000
ei, ej = S.counter_graph_edges(n, m, 1)
rank, k = E.graph_colour_major_order(n, ei, ej, 1)
ei, ej = S.counter_graph_edges(n, m, 1, rank)
part = MG.graph_partition(n, ei, ej, 8, method="builtin")
p = E.Plan(S.mrf_model(n, 16, ei, ej, None, device_const=True, device_dual=True))
r, kk = p.suggest_order(0)
print("ASan+UBSan: G(200000, 1000000): colours", k, "cut", round(float((part[ei] != part[ej]).mean()), 3), "levels", [p.schedule_info(d, 0)["n_levels"] for d in (0, 1)], "pass", p.pass_schedule_info(0)["n_levels"], "suggested colours", kk)
PY
echo "## ThreadSanitizer (planning threads: LPMP_PLAN_THREADS=8)"
( cd lp_mp_amd/csrc && hipcc $FLAGS -fsanitize=thread -o ../../build/exp/liblpmp_engine_tsan.so $SRC ) 2>&1 | tail -3
LD_PRELOAD=$RT/libclang_rt.tsan-x86_64.so LPMP_PLAN_THREADS=8 LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_tsan.so \
  timeout 3000 python -m pytest tests/test_plan_host.py tests/test_graph_host.py tests/test_lockstep.py tests/test_overlap.py tests/test_lp_mirror.py tests/test_oracle_ref.py -q -m "not gpu" -p no:cacheprovider \
  -k "not gloo and not two_process and not four_process and not three_process" > build/exp/tsan_pytest.log 2>&1
grep -v "Gloo\|^$" build/exp/tsan_pytest.log | tail -4
echo "ThreadSanitizer warnings: $(grep -c 'WARNING: ThreadSanitizer' build/exp/tsan_pytest.log)"
LD_PRELOAD=$RT/libclang_rt.tsan-x86_64.so LPMP_PLAN_THREADS=8 LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_tsan.so python - <<'PY' 2>&1 | tail -8
import numpy as np
from lp_mp_amd import engine as E, synthetic as S, multi_gpu as MG
n, m = 200000, 1000000
ei, ej = S.counter_graph_edges(n, m, 1)
rank, k = E.graph_colour_major_order(n, ei, ej, 1)
ei, ej = S.counter_graph_edges(n, m, 1, rank)
part = E.graph_refine_partition(n, ei, ej, (np.arange(n) * 8) // n, 8, 10, 0.03, 0)
p = E.Plan(S.mrf_model(n, 16, ei, ej, None, device_const=True, device_dual=True))
print("TSan: G(200000, 1000000): colours", k, "levels", [p.schedule_info(d, 0)["n_levels"] for d in (0, 1)], "pass", p.pass_schedule_info(0)["n_levels"], "suggested colours", p.suggest_order(0)[1])
g = S.grid_model(256, 256, 32, order="colour_major", device_const=True)
q = E.Plan(g); print("TSan: 256 x 256 x 32 grid: pass rotates", q.pass_rotates(0), q.pass_schedule_info(0)["n_levels"])
PY
} > "$OUT" 2>&1
cat "$OUT"
