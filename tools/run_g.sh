mkdir -p gpurun_out/g
python -m pytest tests/test_b16_gpu.py -x -q > gpurun_out/g/t_b16.log 2>&1; tail -6 gpurun_out/g/t_b16.log
python tools/bench_b16.py --set 3h16=0 --ab 3h16=1 > gpurun_out/g/b16_ab.log 2>&1; cat gpurun_out/g/b16_ab.log | grep "k3 s1"
