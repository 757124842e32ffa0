"""one-off wider fuzz run on the GPU box: python tests/fuzz_more.py FIRST COUNT  (same checks as test_fuzz_gpu.py, other seeds)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_fuzz_gpu as T

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    for fn in (T.test_random_models_all_modes_and_custom_passes, T.test_random_mrfs_fast_kernels_multi_pass_calls_and_fused_custom_schedules,
               T.test_random_mrfs_any_label_count_runtime_dims_kernels, T.test_random_mrfs_primal_rounding):
        try:
            fn(seed)
        except Exception:
            bad += 1
            print("FAIL", fn.__name__, seed)
            traceback.print_exc(limit=3)
print("done", count, "seeds x 4 tests,", bad, "failures")
sys.exit(1 if bad else 0)
