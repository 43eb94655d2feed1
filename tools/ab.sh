set -e
for z in 0 2; do echo "FEATURE_STORE_MODE=$z (bit0 zero-first, bit1 write-through)"; GOENV_ZERO_FIRST=$z GOENV_WAVES_PER_BOARD=3 timeout -k 10 200 python -u tools/tune_waves.py 512 1024 1536 2048 3072 4096 2>&1 | grep " 3 " ; done
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2
timeout -k 10 200 python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err; tail -c 1500 gpurun_out/bench4.json
