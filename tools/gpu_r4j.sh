#!/bin/bash
set -u
export TMPDIR=/tmp
for m in none pdist topk copy; do timeout 600 python tools/stress_one_proc.py 300 $m 2>&1 | grep -E "one process"; done
timeout 600 python tools/stress_one_proc.py 300 topk 64 2>&1 | grep -E "one process"
timeout 600 python tools/stress_one_proc.py 300 topk 600 2>&1 | grep -E "one process"
