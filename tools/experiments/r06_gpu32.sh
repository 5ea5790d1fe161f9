#!/bin/bash
# r06 call 32: the abort of test_gpu_fuzz_seeds under the host-ASan build: the sanitizer's report (pytest -s: fd-level capture swallows it otherwise)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_hostasan/liblambdaworks_kzg.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:handle_segv=0:allocator_may_return_null=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
LD_PRELOAD=$ASAN timeout 600 python -m pytest tests/test_gpu_fuzz_seeds.py -x -q -s -p no:cacheprovider -p no:faulthandler > $O/g32_shipped.txt 2>&1; echo "shipped rc=$?"
grep -v "^  File\|^Extension" $O/g32_shipped.txt | head -60 | cut -c1-300
