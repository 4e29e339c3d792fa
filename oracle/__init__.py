"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the CWSL_DIGI hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  See oracle/cwsl_oracle.h.
"""
