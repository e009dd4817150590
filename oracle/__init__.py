"""CPU oracle package -- test infrastructure only (see gp_oracle.py header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Parity status: PINNED (bit-exact against reference-generated
golden vectors, tests/test_oracle_golden.py).
"""
