#!/bin/bash
# First contact with a multi-GPU node as one JSON record (tools/first_multi_gpu.py); DRY=1: two ranks on one device.
cd "$(dirname "$0")/../.." && exec python tools/first_multi_gpu.py "$@"
