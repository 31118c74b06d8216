"""Framebuffer sharding across the GPUs of one node, one process per GPU (torchrun-style launchers).

The data path is the C ABI's: every rank renders its tiles (tile t belongs to rank t % world) and `rt_render_gather`
(csrc/rt_multi.cpp) moves the shards to rank 0 with ONE grouped ncclSend/ncclRecv exchange — RCCL called from the library,
no torch in it — and puts the tiles in place on rank 0's device. The launcher's own channel (torch.distributed, any backend:
gloo will do) only carries the 128-byte RCCL id from rank 0 to the other ranks: `init_comm`.

The path has no other exchange step: pixels are independent (main.rs:731-784 carries no cross-pixel state), the scene is
replicated. A single-process host drives all GPUs through `api.MultiContext` (rt_render_multi) instead.

`render_sharded` is the older route for callers that already hold a torch process group and want the shards as torch tensors:
the same shard layout, gathered with torch.distributed; `assemble` = rt_untile on the host."""
import numpy as np

from . import _abi as A
from . import api


def shard_params(base, rank, world, tile_size=32):
    """RtParams for `rank` of `world` (same image, same seed: the picture does not depend on world)."""
    return api.make_params(base.width, base.height, base.samples_per_pixel, base.max_depth, base.seed, base.nan_policy, base.flags,
                           tile_size, rank, world, base.pool_slots, base.tail_paths)


def shard_floats(base, world, tile_size=32):
    """Length of every rank's gather buffer = shard 0's size (shards differ by at most one tile)."""
    return api.output_floats(shard_params(base, 0, world, tile_size))


def init_comm(ctx, rank, world, dist=None):
    """Attach an RCCL communicator to `ctx` (rt_comm_init_rank, collective). `dist` = an initialised torch.distributed
    module (or anything with broadcast_object_list) used ONLY to hand rank 0's id to the other ranks."""
    ids = [api.comm_unique_id() if rank == 0 else None]
    if world > 1:
        if dist is None:
            raise ValueError("world > 1 needs a channel for the RCCL id")
        dist.broadcast_object_list(ids, src=0)
    ctx.comm_init_rank(ids[0], rank, world)
    return ids[0]


def render_gathered(ctx, scene, cam, base, rank, output_kind=A.RT_OUT_RGB_SUM_F32, device=None, tile_size=32):
    """rt_render_gather on this rank. Rank 0 returns (frame tensor (H, W, 3) on `device`: float32 sums or uint8, stats);
    the other ranks return (None, stats)."""
    import torch
    prm = shard_params(base, 0, 1, tile_size)     # shard fields are ignored by rt_render_gather (the communicator's rank/world count)
    frame = None
    if rank == 0:
        frame = torch.empty((base.height, base.width, 3), dtype=torch.uint8 if output_kind == A.RT_OUT_RGB8 else torch.float32, device=device)
        if frame.is_cuda:
            torch.cuda.current_stream(frame.device).synchronize()   # the allocation is ordered before the library's stream touches it
    st = ctx.render_gather(scene, cam, prm, output_kind, frame.data_ptr() if frame is not None else None)
    return frame, st


def init_comm_guarded(ctx, rank, world, dist, timeout_s=180.0):
    """init_comm + rt_comm_selftest on a helper thread with a deadline; every rank learns whether ALL ranks got their communicator.
    Returns (ok, reason). A launcher uses it to fall back to render_gathered_staged instead of dying on a node whose RCCL set-up fails
    (a hung set-up leaves the helper thread behind; the process still ends through the launcher)."""
    import threading
    import torch
    err = []

    def work():
        try:
            init_comm(ctx, rank, world, dist)
            ctx.comm_selftest()
        except Exception as e:      # noqa: BLE001 — reported, not swallowed
            err.append(repr(e))
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    mine = "" if (not t.is_alive() and not err) else (err[0] if err else f"RCCL set-up did not finish within {timeout_s:.0f} s")
    flag = torch.tensor([0.0 if mine else 1.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    reasons = [None] * world
    dist.all_gather_object(reasons, mine)
    ok = bool(flag.item() == 1.0)
    return ok, "" if ok else "; ".join(f"rank {r}: {m}" for r, m in enumerate(reasons) if m)


def trial_gather(ctx, scene, cam, rank, world, dist, device, width=160, height=96, spp=2, timeout_s=120.0):
    """One small rt_render_gather through the freshly made communicator, on a helper thread with a deadline; rank 0 renders the same
    frame alone and compares bit for bit (the picture does not depend on the number of shards). Every rank learns the outcome. This is
    the first time the exchange runs between the devices of THIS node, so a launcher calls it before it trusts rt_render_gather."""
    import threading
    import torch
    base = api.make_params(width, height, spp, max_depth=50, seed=1)
    box = {}

    def work():
        try:
            frame, _ = render_gathered(ctx, scene, cam, base, rank, A.RT_OUT_RGB_SUM_F32, device=device)
            box["frame"] = frame
        except Exception as e:      # noqa: BLE001
            box["err"] = repr(e)
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(timeout_s)
    mine = ""
    if t.is_alive():
        mine = f"rt_render_gather did not return within {timeout_s:.0f} s"
    elif "err" in box:
        mine = box["err"]
    elif rank == 0:
        own, _ = ctx.render(scene, cam, base)
        if not np.array_equal(box["frame"].cpu().numpy(), own):
            mine = "gathered frame differs from rank 0's own render of the same frame"
    flag = torch.tensor([0.0 if mine else 1.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    reasons = [None] * world
    dist.all_gather_object(reasons, mine)
    ok = bool(flag.item() == 1.0)
    return ok, "" if ok else "; ".join(f"rank {r}: {m}" for r, m in enumerate(reasons) if m)


def render_gathered_staged(ctx, scene, cam, base, rank, world, dist, output_kind=A.RT_OUT_RGB_SUM_F32, device=None, tile_size=32):
    """The same frame as render_gathered without RCCL: every rank renders its shard (rt_render_device), the shards travel through host
    memory over `dist` (gloo), rank 0 puts the tiles in place on its device (rt_untile_device). RGB8: write_color per shard first
    (rt_resolve_device on the shard buffer — it is per pixel, so tile order does not matter). Slow path for nodes where the library's
    own exchange cannot be set up."""
    import torch
    prm = shard_params(base, rank, world, tile_size)
    n = shard_floats(base, world, tile_size)
    out = torch.zeros(n, dtype=torch.float32, device=device)
    torch.cuda.current_stream(out.device).synchronize()
    st = ctx.render_device(scene, cam, prm, out.data_ptr())
    if output_kind == A.RT_OUT_RGB8:
        out8 = torch.empty(n, dtype=torch.uint8, device=device)
        torch.cuda.current_stream(out.device).synchronize()
        ctx.resolve_device(out.data_ptr(), n // 3, 1, base.samples_per_pixel, out8.data_ptr())
        out = out8
    host = out.cpu()
    parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
    dist.gather(host, gather_list=parts, dst=0)
    if rank != 0:
        return None, st
    gathered = torch.cat(parts).to(device)
    frame = torch.empty((base.height, base.width, 3), dtype=out.dtype, device=device)
    torch.cuda.current_stream(frame.device).synchronize()
    ctx.untile_device(shard_params(base, 0, world, tile_size), output_kind, gathered.data_ptr(), frame.data_ptr())
    return frame, st


def render_sharded(render_shard, base, rank, world, dist=None, tile_size=32, device=None):
    """Render this rank's tiles and gather all shards on rank 0 with torch.distributed.

    render_shard(params, out_tensor) fills out_tensor (1-D float32 torch tensor on `device`, zero
    padded) with this rank's tiles; on the GPU path it calls Context.render_device with
    out_tensor.data_ptr().  Returns the gathered (world, n) tensor on rank 0, None elsewhere."""
    import torch
    prm = shard_params(base, rank, world, tile_size)
    n = shard_floats(base, world, tile_size)
    out = torch.zeros(n, dtype=torch.float32, device=device)
    if out.is_cuda:
        torch.cuda.current_stream(out.device).synchronize()   # torch's fill is done before the library's own stream writes the buffer
    render_shard(prm, out)
    if world == 1 or dist is None:
        return out.unsqueeze(0)
    if rank == 0:
        parts = [torch.empty_like(out) for _ in range(world)]
        dist.gather(out, gather_list=parts, dst=0)
        return torch.stack(parts)
    dist.gather(out, gather_list=None, dst=0)
    return None


def assemble(base, gathered, world, tile_size=32):
    """rt_untile on the host: (world, n) gathered shards -> (H, W, 3) rgb_sum."""
    prm = shard_params(base, 0, world, tile_size)
    return api.untile(prm, np.ascontiguousarray(gathered, dtype=np.float32).reshape(-1))
