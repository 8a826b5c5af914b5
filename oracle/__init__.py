"""TEST INFRASTRUCTURE ONLY -- see oracle/spmv_oracle.c.  Nothing under lilac-benchmarks_amd/
imports this package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
