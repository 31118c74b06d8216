"""Framebuffer sharding across the GPUs of one node: one process per GPU, tiles dealt round-robin
(tile t belongs to rank t % world), one RCCL gather of equal-sized shard buffers to rank 0.

The path has no other exchange step: pixels are independent (main.rs:731-784 carries no cross-pixel
state), the scene is replicated. torch / torch.distributed are plumbing here (device memory and the
collective); the render itself is the C ABI's rt_render_device."""
import numpy as np

from . import api


def shard_params(base, rank, world, tile_size=32):
    """RtParams for `rank` of `world` (same image, same seed: the picture does not depend on world)."""
    return api.make_params(base.width, base.height, base.samples_per_pixel, base.max_depth, base.seed, base.nan_policy, base.flags,
                           tile_size, rank, world, base.pool_slots)


def shard_floats(base, world, tile_size=32):
    """Length of every rank's gather buffer = shard 0's size (shards differ by at most one tile)."""
    return api.output_floats(shard_params(base, 0, world, tile_size))


def render_sharded(render_shard, base, rank, world, dist=None, tile_size=32, device=None):
    """Render this rank's tiles and gather all shards on rank 0.

    render_shard(params, out_tensor) fills out_tensor (1-D float32 torch tensor on `device`, zero
    padded) with this rank's tiles; on the GPU path it calls Context.render_device with
    out_tensor.data_ptr().  Returns the gathered (world, n) tensor on rank 0, None elsewhere."""
    import torch
    prm = shard_params(base, rank, world, tile_size)
    n = shard_floats(base, world, tile_size)
    out = torch.zeros(n, dtype=torch.float32, device=device)
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
