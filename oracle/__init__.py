"""CPU oracle for the needle DFA table-walk path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package (needle_amd/) never does.  See oracle/README.md.
"""
