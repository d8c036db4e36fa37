B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --alt-steps 0 --profile-steps 0"
for q in default 8 16 2 1; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  timeout -k 10 200 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('queues', '$q', d['ms_per_step'], d['value'])" || exit 1
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout -k 10 200 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('packet_capture 0', d['ms_per_step'], d['value'])"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout -k 10 200 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('packet_capture 1', d['ms_per_step'], d['value'])"
