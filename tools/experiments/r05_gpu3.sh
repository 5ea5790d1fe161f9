#!/bin/bash
# host-side finishing + auto rows-per-quad: single-blob API timing, small batches; parity spot check
python tools/single_blob_timing.py 2>&1
LWKZG_DIRECT=13 python tools/single_blob_timing.py 2>&1 | head -3
LWKZG_HOST_FINISH=0 python tools/single_blob_timing.py 2>&1 | head -1
python tools/host_api_timing.py 2>&1 | head -8
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lagrange.py -x -q -m gpu -k "not 1024 and not 4096" 2>&1 | tail -3
