"""CPU oracle for the lvc hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker; the product (lvc_amd/) never does.  See oracle/oracle.c for the
parity-pin statement of each native function and DESIGN.md section "Oracle".
"""
