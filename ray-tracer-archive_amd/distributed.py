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


class CollectiveTimeout(RuntimeError):
    """A rank did not come back from an RCCL call within its deadline: the process still holds a wedged GPU stream and a helper thread
    inside the library, so nothing more can run in it."""


def _die_on_timeout(reason, code=3):
    """Default reaction to a timeout inside a guarded collective: say why and END THE PROCESS, non-zero. A hung ncclRecv sits on the
    context's stream (every later launch on it would queue behind it for ever) and the helper thread is still inside the library on a
    context that is not thread-safe: a fresh process is the only safe retry (the launcher's job). A REFUSAL — the helper came back with an
    error — is different: nothing is in flight, and the caller may fall back to render_gathered_staged in the same process."""
    import os
    import sys
    print("FATAL: " + reason + " — ending this process (a hung collective cannot be recovered from in-process)", file=sys.stderr, flush=True)
    os._exit(code)


def _agree(dist, world, mine, timed_out, on_timeout):
    """Every rank learns every rank's outcome (`mine` = '' or a reason; `timed_out` = this rank's helper thread is still running).
    Returns (ok, reasons joined). If ANY rank timed out, every rank ends through `on_timeout` (default: _die_on_timeout)."""
    outcomes = [None] * world
    dist.all_gather_object(outcomes, (mine, bool(timed_out)))
    ok = all(not m for m, _ in outcomes)
    why = "" if ok else "; ".join(f"rank {r}: {m}" for r, (m, _) in enumerate(outcomes) if m)
    if any(t for _, t in outcomes):
        on_timeout(why)
        raise CollectiveTimeout(why)          # only reached when a test's on_timeout returns
    return ok, why


def init_comm_guarded(ctx, rank, world, dist, timeout_s=180.0, on_timeout=_die_on_timeout):
    """init_comm + rt_comm_selftest on a helper thread with a deadline; every rank learns whether ALL ranks got their communicator.
    Returns (ok, reason) when every rank came back: ok False = RCCL REFUSED somewhere (nothing in flight; a launcher may fall back to
    render_gathered_staged). A rank that did NOT come back within the deadline ends every rank's process (see _die_on_timeout)."""
    import threading
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
    timed_out = t.is_alive()
    mine = f"RCCL set-up did not finish within {timeout_s:.0f} s" if timed_out else (err[0] if err else "")
    return _agree(dist, world, mine, timed_out, on_timeout)


def trial_gather(ctx, scene, cam, rank, world, dist, device, width=160, height=96, spp=2, timeout_s=120.0, on_timeout=_die_on_timeout):
    """One small rt_render_gather through the freshly made communicator, on a helper thread with a deadline; rank 0 renders the same
    frame alone and compares bit for bit (the picture does not depend on the number of shards). Every rank learns the outcome. This is
    the first time the exchange runs between the devices of THIS node, so a launcher calls it before it trusts rt_render_gather.
    (ok, reason) as init_comm_guarded: a refusal or a wrong frame comes back as ok False, a timeout ends the processes."""
    import threading
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
    mine, timed_out = "", t.is_alive()
    if timed_out:
        mine = f"rt_render_gather did not return within {timeout_s:.0f} s"
    elif "err" in box:
        mine = box["err"]
    elif rank == 0:
        own, _ = ctx.render(scene, cam, base)
        if not np.array_equal(box["frame"].cpu().numpy(), own):
            mine = "gathered frame differs from rank 0's own render of the same frame"
    return _agree(dist, world, mine, timed_out, on_timeout)


def render_gathered_staged(ctx, scene, cam, base, rank, world, dist, output_kind=A.RT_OUT_RGB_SUM_F32, device=None, tile_size=32):
    """The same frame as render_gathered without RCCL: every rank renders its shard (rt_render_device), the shards travel through host
    memory over `dist` (gloo), rank 0 puts the tiles in place on its device (rt_untile_device). RGB8: write_color per shard first
    (rt_resolve_device on the shard buffer — it is per pixel, so tile order does not matter). Slow path for nodes where the library's
    own exchange cannot be set up.

    Like rt_render_gather, the ranks AGREE before anything is gathered: a rank whose render failed still takes part, and every rank
    raises — the failed one its own error, the others RtError(RT_ERR_PEER) naming it — instead of rank 0 waiting in the gather."""
    import torch
    prm = shard_params(base, rank, world, tile_size)
    n = shard_floats(base, world, tile_size)
    mine, st, out = None, None, None
    try:
        out = torch.zeros(n, dtype=torch.float32, device=device)
        if out.is_cuda:
            torch.cuda.current_stream(out.device).synchronize()
        st = ctx.render_device(scene, cam, prm, out.data_ptr())
        if output_kind == A.RT_OUT_RGB8:
            out8 = torch.empty(n, dtype=torch.uint8, device=device)
            if out.is_cuda:
                torch.cuda.current_stream(out.device).synchronize()
            ctx.resolve_device(out.data_ptr(), n // 3, 1, base.samples_per_pixel, out8.data_ptr())
            out = out8
    except Exception as e:      # noqa: BLE001 — agreed on below, then re-raised
        mine = e
    outcomes = [None] * world
    dist.all_gather_object(outcomes, None if mine is None else repr(mine))
    failed = [r for r, m in enumerate(outcomes) if m is not None]
    if failed:
        if mine is not None:
            raise mine
        raise api.RtError(A.RT_ERR_PEER, f"rank {failed[0]} failed its part of the render ({outcomes[failed[0]]}): the gather was called off on every rank")
    host = out.cpu()
    parts = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
    dist.gather(host, gather_list=parts, dst=0)
    if rank != 0:
        return None, st
    gathered = torch.cat(parts).to(device)
    frame = torch.empty((base.height, base.width, 3), dtype=out.dtype, device=device)
    if frame.is_cuda:
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
