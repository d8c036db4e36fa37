mkdir -p gpurun_out/f
python tools/bench_bn.py Bpc=0 > gpurun_out/f/bn_old.log 2>&1; tail -13 gpurun_out/f/bn_old.log
python tools/bench_bn.py Bpc=1 > gpurun_out/f/bn_new.log 2>&1; tail -13 gpurun_out/f/bn_new.log
python tools/bench_bn.py Bpc=0 > gpurun_out/f/bn_old2.log 2>&1; tail -1 gpurun_out/f/bn_old2.log
python -m pytest tests/test_ops_gpu.py -x -q -k "bn or scale_act or batch or norm or loader_side" > gpurun_out/f/t.log 2>&1; tail -3 gpurun_out/f/t.log
