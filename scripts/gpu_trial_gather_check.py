"""torchrun with ONE rank: the guarded communicator set-up and the trial gather (distributed.trial_gather) on a world of 1 — the code
bench.py runs first on a multi-GPU node. usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 scripts/gpu_trial_gather_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from importlib import import_module
import rta
pkg = rta.load()
D = import_module("ray_tracer_archive_amd.distributed")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = pkg.Context(0, stream.cuda_stream)
ok, why = D.init_comm_guarded(ctx, rank, world, dist, timeout_s=60.0)
hs = pkg.HostScene("book1", 1)
scene = ctx.upload(hs.desc)
ok2, why2 = D.trial_gather(ctx, scene, hs.camera(160 / 96), rank, world, dist, dev) if ok else (False, "no communicator")
print("world", world, "communicator:", ok, why, "| trial gather:", ok2, why2, flush=True)
assert ok and ok2
dist.destroy_process_group()
