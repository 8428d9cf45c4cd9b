"""CPU oracle for the PlanNextMap path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (blance_amd/) never does.
"""
