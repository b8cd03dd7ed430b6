"""World-size-2 gloo test (CPU) of the N > 1 frame path: tile ownership, the single gather, untile."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, res, name, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    import torch.distributed as dist
    import oracle as orc
    from conftest import oracle_bvh
    from openclraytracer_amd import frame_tiles as ft
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        osc, b = oracle_bvh(name)
        cam = osc.demo_camera(*res)
        # every rank holds the replicated scene; the checker computes the frame, the rank keeps only its tiles
        c, _ = b.trace_primary(cam, nthreads=2)
        frame = orc.shade_frame(osc, cam, c, nthreads=2)
        til = ft.tiling(world, 1)                          # rank 0 owns two tiles of every three (bench.py's default for N > 1)
        slots0 = ft.owned_slots(res[0], res[1], 0, til)
        mine = ft.owned_slots(res[0], res[1], rank, til)
        send = np.zeros((slots0, 4), dtype=np.float32)
        send[:mine] = ft.compact_from_frame(frame, res[0], res[1], rank, til, mine)
        # bench.py's pattern: the gather is issued asynchronously (the next frame's kernels overlap it on the GPU) and
        # waited for before the frame is untiled; the synchronous form must give the same buffer
        out, work_handle = ft.gather_to_root(torch.from_numpy(send), world, rank, async_op=True)
        work_handle.wait()
        out_sync = ft.gather_to_root(torch.from_numpy(send), world, rank)
        if rank == 0:
            assert torch.equal(out, out_sync)
        # the 12-byte wire format bench.py gathers: three planes per rank, hit flag in the sign bit of red
        planes = np.zeros((3, slots0), dtype=np.float32)
        planes[:, :mine] = ft.pack_planes(send[:mine])
        outp = ft.gather_to_root(torch.from_numpy(planes), world, rank)
        if rank == 0:
            got = ft.untile_host(out.numpy(), res[0], res[1], til)
            gotp = ft.untile_host_packed(outp.numpy(), res[0], res[1], til)
            ret['equal'] = bool(np.array_equal(got, frame)) and gotp.tobytes() == frame.tobytes()
            ret['hits'] = int((frame[:, 3] > 0).sum())
        # round 3: another rank assembles the frame (rt_multi_frame_create_rooted; bench.py: scene i gathers into rank i mod N) --
        # the tiles are dealt by virtual rank (rank - root) mod N, the gather lands on `root`
        root = 1
        v = ft.virtual_rank(rank, root, world)
        mine_v = ft.owned_slots(res[0], res[1], v, til)
        planes_v = np.zeros((3, slots0), dtype=np.float32)
        planes_v[:, :mine_v] = ft.pack_planes(ft.compact_from_frame(frame, res[0], res[1], v, til, mine_v))
        outr = ft.gather_to_root(torch.from_numpy(planes_v), world, rank, root=root)
        if rank == root:
            gotr = ft.untile_host_packed(ft.by_virtual_rank(outr.numpy(), root), res[0], res[1], til)
            ret['equal_rooted'] = gotr.tobytes() == frame.tobytes()
        else:
            assert outr is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('res', [(200, 150), (128, 64)])
def test_two_rank_gather_reassembles_the_frame(res):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    procs = [mp.get_context('spawn').Process(target=_worker, args=(r, world, port, res, 'gardenlight', ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs)
    assert ret['equal'] and ret['hits'] > 0 and ret['equal_rooted']


def test_slot_contract_matches_c_abi():
    from openclraytracer_amd import capi, frame_tiles as ft
    for (w, h, ranks, extra) in [(1920, 1080, 8, 0), (1920, 1080, 4, 0), (640, 480, 2, 0), (63, 65, 3, 0),
                                 (1920, 1080, 2, 1), (1920, 1080, 4, 1), (1920, 1080, 8, 1), (3840, 2160, 8, 3), (333, 200, 3, 5)]:
        n = ft.tiling(ranks, extra)                       # == RT_TILING: rank 0 owns `extra` more tiles per period
        assert n == capi.tiling(ranks, extra)
        cam = np.zeros(1, dtype=capi.CAMERA)
        cam['resX'], cam['resY'] = w, h
        for r in range(ranks):
            assert capi.tiles_owned_pixels(cam, r, n) == ft.owned_slots(w, h, r, n)
        a = capi.tile_slot_map(w, h, n)
        b = ft.slot_map(w, h, n)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        # every pixel has exactly one (rank, slot), slots of a rank are distinct and below its count
        for r in range(ranks):
            mine = b[1][b[0] == r]
            assert len(np.unique(mine)) == len(mine) and (len(mine) == 0 or mine.max() < ft.owned_slots(w, h, r, n))
        assert ft.owned_slots(w, h, 0, n) == max(ft.owned_slots(w, h, r, n) for r in range(ranks))   # rank 0 sizes the gather
