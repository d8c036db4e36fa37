set -e
python tools/bench_convs.py --iters 5 > gpurun_out/convs_occ3.txt 2>&1
tail -42 gpurun_out/convs_occ3.txt
python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -2
