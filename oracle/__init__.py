"""TEST INFRASTRUCTURE ONLY — CPU oracle of the PEPPER hot path.

Nothing under oracle/ is product code: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it, and only as the checker / CPU baseline. See oracle/README.md.
"""
