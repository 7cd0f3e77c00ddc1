#!/bin/bash
echo "== plan tests"; timeout 900 python -m pytest tests/test_plan_gpu.py -x -q -m gpu 2>&1 | tail -15
