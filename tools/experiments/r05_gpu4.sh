#!/bin/bash
# host finishing on the proof paths + validation beside the GPU work: single-blob API timing; proof parity
python tools/single_blob_timing.py 2>&1
LWKZG_DIRECT=13 python tools/single_blob_timing.py 2>&1 | head -3
python tools/host_api_timing.py 2>&1 | head -3
timeout 900 python -m pytest tests/test_gpu_proof_parity.py tests/test_gpu_fuzz_seeds.py tests/test_gpu_host_api_extras.py -x -q -m gpu 2>&1 | tail -3
