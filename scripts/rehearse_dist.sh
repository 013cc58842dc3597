# Rehearsal of the N>1 launch path on a 1-GPU box: the driver's torch.distributed.run line with 2
# ranks sharing cuda:0 over gloo (the real run is one rank per GPU over RCCL), then smoke().
set -e
python __graft_entry__.py --smoke
MMC_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 100 --warmup 10 --replicas 2048 --threads 4 --no-cpu --no-secondary
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu --no-secondary
