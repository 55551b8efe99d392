"""TEST INFRASTRUCTURE — CPU oracle package (see oracle/pfd_oracle.c header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
