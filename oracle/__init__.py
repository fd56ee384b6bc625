"""CPU checker for the spot-finder hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (fast-feedback-service_amd/) never does.
"""
