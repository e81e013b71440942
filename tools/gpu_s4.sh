#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -k "pairwise or golden or full_size" 2>&1 | tail -3
timeout 300 python tools/bench_kernels.py pdist 2>&1 | grep "pdist"
