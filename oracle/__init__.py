"""Test infrastructure: CPU restatement of the reference MPPI path (see mppi_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
"""
