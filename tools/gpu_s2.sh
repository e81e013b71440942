#!/bin/bash
export TMPDIR=/tmp
timeout 600 python tools/debug_graph.py 2>&1 | grep -v amdgpu | cut -c1-300
