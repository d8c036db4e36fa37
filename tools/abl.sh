set -e
for L in 256,512,3,1,26 512,256,1,1,52 512,512,3,1,52 128,256,3,1,52; do
python tools/bench_convs.py --only $L --set occ3=0 --ab rpre=0 --ab-default 1 --iters 10 2>&1 | grep "^AB"
done
