#!/bin/bash
# r5q: the new test of long incompressible stretches (all kernel variants)
timeout 600 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "long_incompressible" 2>&1 | grep -v amdgpu.ids | tail -4
