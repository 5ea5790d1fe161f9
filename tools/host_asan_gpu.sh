# The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer on the GPU box (device code built as always:
# make -C lambdaworks_kzg_amd/csrc hostasan): the concurrent host front -- coalesced single-blob callers, proof fronts, two caller
# streams, batch verification, load / free cycles -- with every allocation and every signed overflow of the host side checked.
export LWKZG_EXPERIMENTAL=1   # the A/B arms below are experiment knobs (csrc/knobs.h, r06)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/host_asan
rm -rf $O; mkdir -p $O
cd $R
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
export LWKZG_LIBRARY=$R/lambdaworks_kzg_amd/lib_hostasan/liblambdaworks_kzg.so
# allocator_may_return_null: the test of a table that does not fit makes the HIP runtime's own host allocation fail, which it handles
# reports go to files (log_path): pytest captures stderr at the descriptor, and a process that aborts never prints what was captured --
# r06's first run died on a UBSan report nobody could see
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:handle_segv=0:allocator_may_return_null=1:log_path=$O/san
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1:log_path=$O/san
export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH   # (torch dlopens its own libraries by name: under the sanitizer's dlopen interceptor the RUNPATH of the caller is not consulted)
# the whole GPU suite ($1 = extra pytest arguments, e.g. -k "threads or everything_at_once" for the concurrency tests only)
# (round 4: the three tests that start FRESH python processes whose first HIP call is torch.cuda's own initialisation die inside
# torch's libamdhip64 under the sanitizer's preload -- with the plain library just the same, tools/experiments/r04_gpu18.sh -- and are
# left out here; they run in the plain suite)
# (round 6: the eight-rank rehearsal is the same kind of test as the two-rank one: fresh ranks under torch.distributed.run; the eight-context
# rehearsal runs in a child process that printed its results and then died AT EXIT the way round 5's child did, san.* has the stack)
# (round 5: one more child process, the LWKZG_MODE=ckzg load of test_gpu_zz_env.py, printed the right bytes and then aborted AT EXIT in the
# sanitizer's own device allocator -- sanitizer_allocator_device.h:125, reached from libhsa-runtime64's exit handler; profiles/r05_host_asan_gpu_log.txt)
SKIP="--deselect tests/test_gpu_dist.py::test_two_process_rehearsal_of_a_multi_gpu_job --deselect tests/test_gpu_dist.py::test_rccl_path_at_world_size_1 --deselect tests/test_gpu_dist.py::test_bench_gpus_2_without_a_launcher --deselect tests/test_gpu_dist.py::test_bench_gpus_8_rehearsal_on_one_device --deselect tests/test_gpu_zz_env.py::test_ckzg_mode_from_the_environment_loads_the_lagrange_form --deselect tests/test_gpu_multi.py::test_eight_contexts_on_one_device"
LD_PRELOAD=$ASAN timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider $SKIP $1 > $O/log.txt 2>&1
echo "pytest rc=$?" >> $O/log.txt
for f in $O/san.*; do [ -s "$f" ] && { echo "== sanitizer report $f" >> $O/log.txt; head -40 "$f" | cut -c1-300 >> $O/log.txt; }; done
tail -6 $O/log.txt
grep -c "AddressSanitizer\|runtime error" $O/log.txt
