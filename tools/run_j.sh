mkdir -p gpurun_out/j
python -m pytest tests/test_f8_gpu.py -q > gpurun_out/j/t_f8.log 2>&1; tail -3 gpurun_out/j/t_f8.log | cut -c1-200
python bench.py --precision fp8s --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 2 > gpurun_out/j/bench_fp8s.json 2> gpurun_out/j/bench_fp8s.err; tail -c 1500 gpurun_out/j/bench_fp8s.json; tail -5 gpurun_out/j/bench_fp8s.err
python bench.py --precision bf16s --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > gpurun_out/j/bench_bf16s.json 2> gpurun_out/j/bench_bf16s.err; tail -c 700 gpurun_out/j/bench_bf16s.json
