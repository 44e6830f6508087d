#!/bin/bash
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/full/tests.txt
cat gpurun_out/full/tests.txt
