"""CPU oracle: TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package (as the checker, never as the thing measured or shipped)."""
