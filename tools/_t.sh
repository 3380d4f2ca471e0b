set -e
timeout -k 10 600 python -m pytest tests/test_gpu_snark.py tests/test_gpu_parity.py -m gpu -x -q -k "poly or prover or setup" > gpurun_out/t_ntt.log 2>&1 || { tail -30 gpurun_out/t_ntt.log; exit 1; }
tail -2 gpurun_out/t_ntt.log
timeout -k 10 300 python tools/chain_prof.py 248 5
