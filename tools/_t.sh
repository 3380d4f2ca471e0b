set -e
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1 || { tail -40 gpurun_out/t_full.log; exit 1; }
tail -2 gpurun_out/t_full.log
