"""Under torchrun with 2+ ranks on a ONE-GPU box (every rank on device 0): the staged gather (distributed.render_gathered_staged — the
bench's fallback when the library's RCCL exchange cannot be set up; RCCL refuses two ranks on one device) must give the single-GPU
frame bit for bit, f32 sums and RGB8.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 scripts/gpu_staged_gather_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import numpy as np
from importlib import import_module
import rta
pkg = rta.load()
D = import_module("ray_tracer_archive_amd.distributed")
A = pkg._abi
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = pkg.Context(0, stream.cuda_stream)
ok, why = D.init_comm_guarded(ctx, rank, world, dist, timeout_s=60.0)
hs = pkg.HostScene("book1", 1)
scene = ctx.upload(hs.desc)
W, H, spp = 333, 207, 8                      # clipped edge tiles
cam = hs.camera(W / H)
prm = pkg.make_params(W, H, spp, max_depth=50, seed=1)
f32, _ = D.render_gathered_staged(ctx, scene, cam, prm, rank, world, dist, A.RT_OUT_RGB_SUM_F32, device=dev)
u8, _ = D.render_gathered_staged(ctx, scene, cam, prm, rank, world, dist, A.RT_OUT_RGB8, device=dev)
if rank == 0:
    ref, _ = ctx.render(scene, cam, prm)
    ref8 = pkg.tonemap(ref, spp)
    same32 = bool(np.array_equal(f32.cpu().numpy(), ref)); same8 = bool(np.array_equal(u8.cpu().numpy(), ref8))
    print("world", world, "library exchange set up:", ok, "|", why[:120], "| staged f32 == single-GPU:", same32, "| staged RGB8 == write_color(single-GPU):", same8, flush=True)
    assert same32 and same8
# a failure injected into rank 1's render (rt_test_fail_next_renders): every rank must come back with an error — rank 1 its own, the
# others RT_ERR_PEER naming rank 1 — instead of rank 0 waiting in the gather; and the next frame must be fine again
if world > 1:
    if rank == 1:
        ctx.fail_next_renders(1)
    try:
        D.render_gathered_staged(ctx, scene, cam, prm, rank, world, dist, A.RT_OUT_RGB_SUM_F32, device=dev)
        outcome = "no error"
    except pkg.RtError as e:
        outcome = "own" if e.code == A.RT_ERR_DEVICE and "injected" in str(e) else ("peer1" if e.code == A.RT_ERR_PEER and "rank 1" in str(e) else repr(e))
    got = [None] * world
    dist.all_gather_object(got, outcome)
    again, _ = D.render_gathered_staged(ctx, scene, cam, prm, rank, world, dist, A.RT_OUT_RGB_SUM_F32, device=dev)
    if rank == 0:
        fine = got[1] == "own" and all(g == "peer1" for r, g in enumerate(got) if r != 1) and bool(np.array_equal(again.cpu().numpy(), ref))
        print("injected failure on rank 1:", got, "| every rank returned and the next frame is right:", fine, flush=True)
        assert fine
dist.barrier()
dist.destroy_process_group()
