# Development check of the N > 1 launch path on a 1-GPU box: two ranks under torch.distributed.run, both on device 0.
# RCCL normally refuses two ranks on one device; if it does, the rendezvous / launch part has still been exercised.
cd $GRAFT_REPO_ROOT
export SRX_BENCH_DEVICE=0 NCCL_DEBUG=WARN
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --cells 200000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/two_ranks.log 2>&1; grep -v "^\s*$" gpurun_out/two_ranks.log | head -60
