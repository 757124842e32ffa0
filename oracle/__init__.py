"""CPU oracle for the LP_MP sweep — TEST INFRASTRUCTURE ONLY (see oracle/lpmp_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
