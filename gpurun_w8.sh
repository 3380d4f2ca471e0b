set -e
timeout -k 10 600 python -m pytest tests/test_gpu_evalmm.py -m gpu -x -q -k "witness" > gpurun_out/t_w8.log 2>&1 || { tail -30 gpurun_out/t_w8.log; exit 1; }
tail -3 gpurun_out/t_w8.log
timeout -k 10 300 python tools/chain_prof.py 248 5
timeout -k 10 600 python -m pytest tests/test_gpu_batch_sizes.py -m gpu -x -q > gpurun_out/t_bs.log 2>&1 || { tail -30 gpurun_out/t_bs.log; exit 1; }
tail -3 gpurun_out/t_bs.log
for per in 124 248; do echo per=$per; MFH_WITNESS_PER=$per timeout -k 10 300 python tools/batch_time.py 992 2>&1 | tail -1; done
