"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms for the hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything from here;
the product (semantic-embeddings_amd/) never does.
"""
