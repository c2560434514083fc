#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python gpurun_micro.py 2>&1 | grep -E "near|1.0h|1.5h|2.5h|replay|loop"
python bench.py --no-cpu-baseline 2>&1 | grep "{" | cut -c1-140
