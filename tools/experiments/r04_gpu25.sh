mkdir -p gpurun_out/r04q
timeout 600 python -m pytest tests/test_gpu_zz_env.py -m gpu -x -q 2>&1 | tail -4
timeout 1500 python tools/soak.py --batches 1200 --direct-bits 16 > gpurun_out/r04q/soak_long.json 2> gpurun_out/r04q/soak_err.txt; echo "{\"soak_rc\": $?}" >> gpurun_out/r04q/soak_long.json
timeout 900 python tools/soak_ckzg.py --batches 300 > gpurun_out/r04q/soak_ckzg_long.json 2>> gpurun_out/r04q/soak_err.txt
tail -2 gpurun_out/r04q/soak_long.json; tail -1 gpurun_out/r04q/soak_ckzg_long.json
