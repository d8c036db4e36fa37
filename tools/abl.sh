set -e
for L in 32,64,3,2,416 32,64,3,1,208 64,32,1,1,208 256,32,1,1,52; do
  python tools/bench_convs.py --only $L --ab split=16 --iters 10 2>&1 | grep "^AB"
done
