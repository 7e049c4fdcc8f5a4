# how a busy host affects the pipeline step (development helper): N busy-loop processes; wait mode spin / yield
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2))'
echo -n "quiet, spin : "; $B 2>/dev/null | python -c "$P"
echo -n "quiet, yield: "; SRX_WAIT=yield $B 2>/dev/null | python -c "$P"
for N in 32; do
  for i in $(seq 1 $N); do (timeout 40 sh -c 'while :; do :; done' &) ; done
  sleep 1
  echo -n "$N spinners, spin : "; $B 2>/dev/null | python -c "$P"
  echo -n "$N spinners, yield: "; SRX_WAIT=yield $B 2>/dev/null | python -c "$P"
  echo -n "$N spinners, spin, no graphs: "; SRX_NO_GRAPH=1 $B 2>/dev/null | python -c "$P"
  sleep 20
done
