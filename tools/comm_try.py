"""Can two ranks share ONE GPU under RCCL on this box?  (If yes, the in-library exchange can be rehearsed on the one-GPU box.)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/comm_try.py"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import torch, torch.distributed as dist
from spaghettisearch_amd import engine, sharding, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
ctx = engine.Context(0)
try:
    sharding.init_lib_comm(ctx, rank, world)
    print(f"rank {rank}: communicator up {ctx.comm_info()}", flush=True)
    n, e = 20000, 100000
    ptr, dst = synth.rmat_graph(n, e, seed=3)
    n_topic = synth.topic_sizes(n, 16)
    g = engine.Graph(ctx, n, ptr, dst, rank=rank, world=world)
    for ar in (False, True):
        ids, r, it = g.pagerank_sharded(0.75, 1e-9, n_topic, allreduce=ar)
        objs = [None] * world
        dist.all_gather_object(objs, (ids, r))
        full = sharding.assemble(objs, n, 16)
        if rank == 0:
            from oracle import pyoracle
            ref, rit = pyoracle.pagerank(n, ptr, dst, 0.75, 1e-9, n_topic)
            print("allreduce" if ar else "allgather", "iters equal", it.tolist() == rit.tolist(), "max rel err", float(np.max(np.abs(full - ref) / ref)), flush=True)
    g.close()
except Exception as exc:
    print(f"rank {rank}: FAILED {exc!r}", flush=True)
ctx.close()
dist.destroy_process_group()
