"""CPU oracle for the MultiGrid step path — TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
`oracle.c_oracle` wraps oracle/mg_oracle.c (literal C restatement); `oracle.py_ref_env` is an
independent pure-Python object-style restatement used for cross-checking and CPU timing.
Parity status: see the header of mg_oracle.c ("parity unpinned" except MiniGridEnv._reward).
"""
