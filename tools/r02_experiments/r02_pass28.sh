#!/bin/bash
# table rows aligned to 128-byte lines when they leave headroom: full GPU suite, the default bench line, free memory
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/p28; mkdir -p $O
python3 -c "import torch; f,t=torch.cuda.mem_get_info(); print('free %.2f GB of %.2f GB (%.2f GiB)' % (f/1e9, t/1e9, t/2**30))" > $O/mem.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python3 - <<'PY' >> $O/tests.txt
import json
j = json.load(open("gpurun_out/p28/bench.json"))
print("headline", round(j["value"]), j["msm_path"], "| build s", j.get("direct_table_build_s"))
for k in ("default_engine", "bucket_engine", "host_abi"):
    e = j[k]; print(k, round(e["value"]), e.get("table_row_bytes"), e.get("table_bytes"), e.get("setup_load_s_incl_table_build"))
PY
cat $O/mem.txt $O/tests.txt; tail -2 $O/bench.err
