"""Oracle = CPU restatement of the reference algorithm. TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
by the product path (cra5_amd/)."""
