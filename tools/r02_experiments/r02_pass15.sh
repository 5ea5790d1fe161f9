#!/bin/bash
# full GPU suite on HEAD + coalesced point-proof rate + a bench sanity line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/p15
timeout 900 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "passed|failed|error|coalesced|Error|assert" | tail -30 > gpurun_out/p15/tests.txt
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/p15/bench.json 2> gpurun_out/p15/bench.err
tail -5 gpurun_out/p15/tests.txt; cat gpurun_out/p15/bench.json | cut -c1-600
