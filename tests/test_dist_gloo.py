"""CPU suite: the N>1 path (shard -> one gather -> assemble on rank 0) with world_size 2
and 3 over gloo.  The shards are packed by the host mirror of the GPU layout; the GPU
tests check that mirror against rt_render_shard / rt_assemble_shards themselves."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    from raytracer_amd import dist as rd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)   # what a full render would give
        shard = torch.from_numpy(rd.pack_shard_host(frame, rank, world))
        allsh = rd.gather_shards(shard, rank, world)
        if rank == 0:
            got = rd.assemble_host(allsh.numpy(), world, w, h)
            q.put(bool(np.array_equal(got, frame)))
        else:
            assert allsh is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 200, 120), (3, 97, 65)])
def test_shard_gather_assemble_gloo(world, w, h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
