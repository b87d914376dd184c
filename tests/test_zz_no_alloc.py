"""GPU suite (-m gpu), collected last: the allocation-free entry point rt_assemble_shards_noalloc against
rt_assemble_shards, through the C-ABI.  The two must give the same bytes; the fixed per-GPU table keeps
the previous call's contents, so the second call uses a SMALLER film and fewer ranks than the first."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import api
from conftest import load_scene

pytestmark = pytest.mark.gpu


def _both(t, scene, world, shards):
    """-> frames of rt_assemble_shards and rt_assemble_shards_noalloc for the same packed shards"""
    import torch
    lib = t._lib
    a = torch.full((scene.height, scene.width, 3), 3, dtype=torch.uint8, device="cuda:0")
    b = torch.full((scene.height, scene.width, 3), 5, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    assert lib.rt_assemble_shards(t._ctx, shards.data_ptr(), world, a.data_ptr()) == api.RT_OK
    assert lib.rt_assemble_shards_noalloc(t._ctx, shards.data_ptr(), world, b.data_ptr()) == api.RT_OK
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_noalloc_assemble_equals_assemble_on_two_calls_in_a_row():
    import torch
    big = load_scene("outdoor_spot").copy(width=333, height=211, spp=1)
    small = load_scene("outdoor_spot").copy(width=70, height=40, spp=1)
    with ra.Tracer() as t:
        # first call: rendered shards of a 333 x 211 film over 5 ranks (7 tile rows x 5 ranks of table)
        t.load_scene(big)
        whole, _, _ = t.render()
        world = 5
        sb = ra.shard_bytes(big.width, big.height, world)
        shards = torch.zeros(world * sb, dtype=torch.uint8, device="cuda:0")
        for r in range(world):
            t.render_shard(r, world, shards.data_ptr() + r * sb)
        a, b = _both(t, big, world, shards)
        assert np.array_equal(a, whole) and np.array_equal(b, a)
        # second call, smaller in every way: 70 x 40 film, 2 ranks, random shard bytes (the call is a permutation)
        t.load_scene(small)
        world = 2
        sb = ra.shard_bytes(small.width, small.height, world)
        shards = torch.randint(0, 256, (world * sb,), dtype=torch.uint8, device="cuda:0")
        a, b = _both(t, small, world, shards)
        assert np.array_equal(b, a)
        from raytracer_amd import dist as rd
        assert np.array_equal(a, rd.assemble_host(shards.cpu().numpy(), world, small.width, small.height))
        # a second context on the same GPU shares the table and must see the same result
        with ra.Tracer() as t2:
            t2.load_scene(small)
            a2, b2 = _both(t2, small, world, shards)
            assert np.array_equal(b2, a)
        # beyond the fixed table: a clear error, nothing launched
        tall = load_scene("outdoor_spot").copy(width=8, height=32 * 4100, spp=1)
        t.load_scene(tall)
        dummy = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
        rc = t._lib.rt_assemble_shards_noalloc(t._ctx, dummy.data_ptr(), 4, dummy.data_ptr())
        assert rc == api.RT_ERR_UNSUPPORTED and b"fixed table" in t._lib.rt_last_error()
