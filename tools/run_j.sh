mkdir -p gpurun_out/j
python -m pytest tests/test_f8_gpu.py -q > gpurun_out/j/t_f8.log 2>&1; tail -3 gpurun_out/j/t_f8.log | cut -c1-200
python -m pytest tests/test_model_gpu.py -q -k "reduced_precision" > gpurun_out/j/t_model.log 2>&1; tail -3 gpurun_out/j/t_model.log | cut -c1-200
python bench.py --precision fp8s --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 2 > gpurun_out/j/bench_fp8s.json 2> gpurun_out/j/bench_fp8s.err; python -c "
import json; d=json.loads(open('gpurun_out/j/bench_fp8s.json').read()); print('fp8s', d['ms_per_step'], d['loss'])"
cp gpurun_out/bench_full_latest.json gpurun_out/j/bench_full_fp8s.json
python bench.py --precision bf16s --steps 10 --warmup 3 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > gpurun_out/j/bench_bf16s.json 2> gpurun_out/j/bench_bf16s.err; python -c "
import json; d=json.loads(open('gpurun_out/j/bench_bf16s.json').read()); print('bf16s', d['ms_per_step'], d['loss'])"
python bench.py --precision fp8s --clips 32 --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > gpurun_out/j/bench_fp8s_c32.json 2> gpurun_out/j/bench_fp8s_c32.err; python -c "
import json; d=json.loads(open('gpurun_out/j/bench_fp8s_c32.json').read()); print('fp8s clips32', d['ms_per_step'], d['value'], d['mem_gb'])"
python bench.py --precision bf16s --clips 32 --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 --profile-steps 0 > gpurun_out/j/bench_bf16s_c32.json 2> gpurun_out/j/bench_bf16s_c32.err; python -c "
import json; d=json.loads(open('gpurun_out/j/bench_bf16s_c32.json').read()); print('bf16s clips32', d['ms_per_step'], d['value'], d['mem_gb'])"
