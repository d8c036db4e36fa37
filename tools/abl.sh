set -e
python tools/bench_convs.py --ab tile64=0 --ab-default 1 --iters 5 > gpurun_out/ab_tile64.txt 2>&1
grep "^AB\|^total\|with" gpurun_out/ab_tile64.txt
