#!/bin/bash
# r06 call 33: the GPU suite and the default bench line on the round's final HEAD (after the sentinel fix), the library built from scratch on the box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
rm -rf lambdaworks_kzg_amd/build lambdaworks_kzg_amd/lib/liblambdaworks_kzg.so
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > $O/g33_gpu_test_log.txt 2>&1; echo "pytest rc=$?" >> $O/g33_gpu_test_log.txt
tail -4 $O/g33_gpu_test_log.txt
LWKZG_BENCH_DETAIL=$O/g33_bench_detail.json python bench.py > $O/g33_bench_line.json 2> $O/g33_bench_err.txt
tail -1 $O/g33_bench_line.json | cut -c1-400
LWKZG_TIMING=1 python tools/verify_device_loop.py --n 4096 --calls 3 --host 2>&1 | grep "staged verification\|verify batch" | tail -4 | tee $O/g33_staged_split.txt
