cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:handle_segv=1:allocator_may_return_null=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH
echo "--- normal lib under the ASan preload"
LD_PRELOAD=$ASAN python tests/dist_gpu_worker.py 0 1 29511 /tmp/out.json gloo 2>&1 | grep -v amdgpu.ids | tail -6
echo "--- asan lib, python -X faulthandler"
export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_hostasan/liblambdaworks_kzg.so
LD_PRELOAD=$ASAN python -X faulthandler tests/dist_gpu_worker.py 0 1 29512 /tmp/out2.json gloo 2>&1 | grep -v amdgpu.ids | grep -A12 "Current thread\|most recent call" | head -30
