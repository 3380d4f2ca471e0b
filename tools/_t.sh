set -e
timeout -k 10 900 python -m pytest tests/test_gpu_prg_ssp.py tests/test_gpu_batch_sharded.py tests/test_gpu_evalmm.py -m gpu -x -q > gpurun_out/t_cols.log 2>&1 || { tail -40 gpurun_out/t_cols.log; exit 1; }
tail -3 gpurun_out/t_cols.log
