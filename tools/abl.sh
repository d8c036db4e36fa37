set -e
for L in 256,512,3,1,52 512,256,1,1,52 512,1024,3,1,26 128,256,3,1,52 256,512,3,1,26; do
  python tools/bench_convs.py --only $L --ab abl=2 --iters 10 2>&1 | grep "^AB"
done
