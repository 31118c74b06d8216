"""The N>1 path (tile sharding + gather + untile) on CPU: world_size 2 and 3, gloo.

The render of a shard needs a GPU, so here each rank fills its shard buffer from a known full frame
through an independent Python statement of the tile layout; what is under test is everything around
it: shard_params / shard_floats / render_sharded (the collective) / assemble (rt_untile)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, W, H, ts, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rta
        pkg = rta.load()
        from importlib import import_module
        D = import_module("ray_tracer_archive_amd.distributed")
        from test_host import py_tile_layout
        full = np.random.default_rng(42).uniform(0, 100, size=(H * W, 3)).astype(np.float32)   # same on every rank
        maps = py_tile_layout(W, H, ts, world)
        base = pkg.make_params(W, H, 4, seed=9)

        def render_shard(prm, out):
            assert prm.shard_index == rank and prm.shard_count == world and prm.tile_size == ts and prm.seed == 9
            m = maps[rank]
            n_local = pkg.output_floats(prm) // 3
            buf = np.zeros((out.numel() // 3, 3), dtype=np.float32)
            sel = m[:n_local] >= 0
            buf[:n_local][sel] = full[m[:n_local][sel]]
            out.copy_(torch.from_numpy(buf.reshape(-1)))

        g = D.render_sharded(render_shard, base, rank, world, dist, tile_size=ts, device="cpu")
        if rank == 0:
            assert g.shape == (world, D.shard_floats(base, world, ts))
            img = D.assemble(base, g.numpy(), world, ts)
            q.put(bool(np.array_equal(img.reshape(-1, 3), full)))
        else:
            assert g is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,ts", [(2, 100, 60, 16), (3, 70, 45, 8)])
def test_sharded_gather_gloo(world, W, H, ts):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, W, H, ts, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True


def id_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rta
        pkg = rta.load()
        from importlib import import_module
        D = import_module("ray_tracer_archive_amd.distributed")
        api = import_module("ray_tracer_archive_amd.api")
        api.comm_unique_id = lambda: bytes(range(128))        # rank 0's id (the real one needs librccl + a GPU)

        class FakeCtx:
            def comm_init_rank(self, uid, r, w):
                self.got = (bytes(uid), r, w)
        c = FakeCtx()
        D.init_comm(c, rank, world, dist)
        q.put(c.got == (bytes(range(128)), rank, world))
    finally:
        dist.destroy_process_group()


def test_rccl_id_travels_over_the_launchers_channel():
    """distributed.init_comm: rank 0's 128-byte id reaches every rank over torch.distributed (gloo here) and each rank attaches with
    its own rank / world. The communicator itself is exercised on the GPU box (tests/test_gpu_multi.py)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True and q.get() is True


def test_untile_rgb8_matches_untile():
    sys.path.insert(0, ROOT)
    import rta
    pkg = rta.load()
    from importlib import import_module
    D = import_module("ray_tracer_archive_amd.distributed")
    for (W, H, ts, world) in [(70, 45, 8, 3), (100, 60, 16, 2), (33, 17, 32, 8)]:
        base = pkg.make_params(W, H, 4)
        n = D.shard_floats(base, world, ts)
        g = np.random.default_rng(W).integers(0, 255, size=(world, n)).astype(np.uint8)
        a = pkg.untile_rgb8(D.shard_params(base, 0, world, ts), g.reshape(-1))
        b = D.assemble(base, g.astype(np.float32), world, ts)
        assert np.array_equal(a.astype(np.float32), b)


def guard_worker(rank, world, port, mode, q):
    """init_comm_guarded / render_gathered_staged with fake contexts: mode 'refuse' (rank 1's RCCL set-up raises), 'hang' (rank 1's set-up
    never returns), 'render' (rank 1's render raises)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        import rta
        pkg = rta.load()
        from importlib import import_module
        D = import_module("ray_tracer_archive_amd.distributed")
        api = import_module("ray_tracer_archive_amd.api")
        A = pkg._abi
        api.comm_unique_id = lambda: bytes(range(128))

        class FakeCtx:
            def comm_init_rank(self, uid, r, w):
                if mode == "refuse" and r == 1:
                    raise api.RtError(A.RT_ERR_DEVICE, "ncclCommInitRank: invalid usage")
                if mode == "hang" and r == 1:
                    time.sleep(3600)

            def comm_selftest(self):
                pass

            def render_device(self, scene, cam, prm, ptr):
                if mode == "render" and rank == 1:
                    raise api.RtError(A.RT_ERR_DEVICE, "injected failure")
                return {}
        c = FakeCtx()
        if mode in ("refuse", "hang"):
            died = []
            try:
                ok, why = D.init_comm_guarded(c, rank, world, dist, timeout_s=2.0, on_timeout=lambda reason: died.append(reason))
                q.put((rank, "returned", ok, why))
            except D.CollectiveTimeout as e:
                q.put((rank, "timeout", bool(died), str(e)))
        else:
            base = pkg.make_params(64, 48, 2)
            try:
                D.render_gathered_staged(c, None, None, base, rank, world, dist, device="cpu")
                q.put((rank, "no error", 0, ""))
            except api.RtError as e:
                q.put((rank, "error", e.code, str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["refuse", "hang", "render"])
def test_guarded_collectives_refusal_timeout_and_failed_rank(mode):
    """(a) RCCL REFUSES on one rank: every rank gets (False, reason) and may fall back in-process. (b) one rank's set-up HANGS: every rank
    takes the timeout exit (os._exit(3) in production; a recording stand-in here) — no in-process fallback over a wedged stream.
    (c) one rank's RENDER fails in the staged gather: every rank raises, the failed one its own error, the others RT_ERR_PEER naming it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=guard_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get()
        got[r[0]] = r[1:]
    for p in procs:
        p.join(30)
        if p.is_alive():          # the 'hang' rank's helper thread sleeps for ever (daemon thread; the process ends with its main thread)
            p.terminate()
    if mode == "refuse":
        assert got[0][0] == got[1][0] == "returned" and got[0][1] is False and got[1][1] is False
        assert "rank 1" in got[0][2] and "invalid usage" in got[0][2] and got[0][2] == got[1][2]
    elif mode == "hang":
        assert got[0][0] == got[1][0] == "timeout" and got[0][1] and got[1][1]
        assert "rank 1" in got[0][2] and "did not finish" in got[0][2]
    else:
        from importlib import import_module
        sys.path.insert(0, ROOT)
        import rta
        A = rta.load()._abi
        assert got[1][0] == "error" and got[1][1] == A.RT_ERR_DEVICE and "injected" in got[1][2]
        assert got[0][0] == "error" and got[0][1] == A.RT_ERR_PEER and "rank 1" in got[0][2]
