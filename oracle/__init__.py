"""CPU restatements of the reference's algorithms (test infrastructure only: tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product path never does)."""
