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
        # a second context on the same GPU has a table slice of its own and must see the same result
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


def test_noalloc_assemble_captured_from_two_contexts_and_replayed_side_by_side():
    """The call under stream capture in GLOBAL mode (any hipMalloc / hipFree in the process would fail the capture), from
    two contexts with different films and rank counts on streams of their own; the two graphs are then replayed
    at the same time, repeatedly, with fresh shard bytes: every context has its own slice of the GPU's table."""
    import torch
    scenes = [load_scene("outdoor_spot").copy(width=333, height=211, spp=1), load_scene("outdoor_spot").copy(width=97, height=300, spp=1)]
    worlds = [5, 3]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    tracers = [ra.Tracer(stream=s.cuda_stream) for s in streams]
    try:
        shards, frames, graphs = [], [], []
        for t, scene, world in zip(tracers, scenes, worlds):
            t.load_scene(scene)
            sb = ra.shard_bytes(scene.width, scene.height, world)
            shards.append(torch.zeros(world * sb, dtype=torch.uint8, device="cuda:0"))
            frames.append(torch.zeros((scene.height, scene.width, 3), dtype=torch.uint8, device="cuda:0"))
        torch.cuda.synchronize()
        for t, s, sh, fr, world in zip(tracers, streams, shards, frames, worlds):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                rc = t._lib.rt_assemble_shards_noalloc(t._ctx, sh.data_ptr(), world, fr.data_ptr())
            assert rc == api.RT_OK, t._lib.rt_last_error()
            graphs.append(g)
        for rep in range(4):
            for sh in shards:
                sh.copy_(torch.randint(0, 256, sh.shape, dtype=torch.uint8, device="cuda:0"))
            for fr in frames:
                fr.zero_()
            torch.cuda.synchronize()
            for g in graphs:
                g.replay()
            torch.cuda.synchronize()
            for t, scene, sh, fr, world in zip(tracers, scenes, shards, frames, worlds):
                want = torch.zeros_like(fr)
                assert t._lib.rt_assemble_shards(t._ctx, sh.data_ptr(), world, want.data_ptr()) == api.RT_OK
                torch.cuda.synchronize()
                assert torch.equal(fr, want), (rep, scene.width)
        del graphs
    finally:
        for t in tracers:
            t.close()
