#!/bin/bash
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -k "cli_end" ) 2>&1 | tail -15
