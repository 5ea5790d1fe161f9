#!/bin/bash
# the entry points the driver calls, then long soaks on the round's final code
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python tools/soak.py --batches 900 --direct-bits 16 2>/dev/null | tail -1 > $O/soak_long.json
timeout 900 python tools/soak.py --batches 300 --direct-bits 0 2>/dev/null | tail -1 >> $O/soak_long.json
timeout 900 python tools/soak_ckzg.py --batches 240 2>/dev/null | tail -1 >> $O/soak_long.json
timeout 600 python tools/soak_small.py --rounds 3000 2>/dev/null | tail -1 >> $O/soak_long.json
LWKZG_DIRECT=16 timeout 700 python tools/soak_verify.py 480 2>/dev/null | tail -1 >> $O/soak_long.json
timeout 700 python tools/soak_verify.py 480 2>/dev/null | tail -1 >> $O/soak_long.json
cat $O/soak_long.json
bash tools/stress_mirror.sh 2>&1 | grep -v amdgpu.ids | tail -4
