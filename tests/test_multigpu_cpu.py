"""N > 1 path on CPU: world_size-2 gloo processes shard the path indices exactly like bench.py does and all-reduce
the framebuffer; the result must equal the single-process render of the union of the indices. The per-rank renderer
here is the oracle (test infrastructure) -- the sharding/reduce logic is what is under test, no GPU needed."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import REPO, SCENE_0010, load_pkg, make_scene, oracle_lib, oracle_pixels, oracle_render, oracle_render_tiles


def _worker(rank, world, port, outdir):
    sys.path.insert(0, str(REPO / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    scene = make_scene(SCENE_0010, width=64, height=64, max_verts=4)
    per_frame = 2 * scene.width * scene.height
    total = np.zeros((scene.height, scene.width, 3), dtype=np.float32)
    shape = (scene.height, scene.width, 3)
    reducer = pkg.FrameReducer([torch.zeros(shape), torch.zeros(shape)], dist)     # bench.py's double-buffered reduce
    frames = 3
    for k in range(frames):                                         # as bench.py's step(k)
        first, count = pkg.shard_range(k * world * per_frame, world * per_frame, rank, world)
        buf = reducer.begin(k)
        assert float(buf.abs().sum()) == 0.0                        # cleared, its previous reduce (frame k-2) is complete
        fb, _, _ = oracle_render(scene, first, count, threads=1)
        buf += torch.from_numpy(fb)
        reducer.end(k)                                              # the framebuffer reduce (RCCL on the GPU box), asynchronous
        if k >= 1:
            total += reducer.finished(k - 1).numpy()                # frame k-1 is complete while frame k's reduce is in flight
    total += reducer.finished(frames - 1).numpy()
    reducer.drain()
    if rank == 0:
        np.save(os.path.join(outdir, "reduced.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    pkg = load_pkg()
    for first, count, world in ((0, 10, 3), (7, 1, 4), (1 << 40, 942080 * 64, 8), (5, 0, 2)):
        parts = [pkg.shard_range(first, count, r, world) for r in range(world)]
        assert parts[0][0] == first and sum(c for _, c in parts) == count
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_ranks_equal_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d), nprocs=world, join=True)
        reduced = np.load(os.path.join(d, "reduced.npy"))
    scene = make_scene(SCENE_0010, width=64, height=64, max_verts=4)
    per_frame = 2 * scene.width * scene.height
    single, _, _ = oracle_render(scene, 0, 3 * world * per_frame, threads=1)
    assert np.allclose(reduced, single, rtol=1e-5, atol=1e-4)
    assert reduced.sum() > 0


def _tile_worker(rank, world, port, outdir):
    """tile-owned sharding (mi_render_tiles / bench.py --shard tiles): rank r renders the 32 x 32 tiles t = r (mod world) of every frame,
    then the same framebuffer all-reduce"""
    sys.path.insert(0, str(REPO / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = load_pkg()
    scene = make_scene(SCENE_0010, width=96, height=64, max_verts=4)
    shape = (scene.height, scene.width, 3)
    reducer = pkg.FrameReducer([torch.zeros(shape), torch.zeros(shape)], dist)
    total = np.zeros(shape, dtype=np.float32)
    steps, spp = 2, 3
    for k in range(steps):
        buf = reducer.begin(k)
        with oracle_pixels():
            fb, cnt, _ = oracle_render_tiles(scene, k * spp, spp, rank, world, threads=1)
        # this rank splats into its own tiles and at most two pixels beyond them (the 4 x 4 filter footprint)
        ty, tx = np.arange(scene.height) // 32, np.arange(scene.width) // 32
        mine = (ty[:, None] * (scene.width // 32) + tx[None, :]) % world == rank
        rim = np.zeros_like(mine)
        for dy in range(-2, 3):
            for dx in range(-2, 3):
                rim |= np.roll(np.roll(mine, dy, axis=0), dx, axis=1)
        assert fb[~rim].sum() == 0.0 and cnt[4] == spp * mine.sum()
        buf += torch.from_numpy(fb)
        reducer.end(k)
        if k >= 1:
            total += reducer.finished(k - 1).numpy()
    total += reducer.finished(steps - 1).numpy()
    reducer.drain()
    if rank == 0:
        np.save(os.path.join(outdir, "reduced_tiles.npy"), total)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_owning_tiles_equal_single_process():
    """... and equal the single-process render of the frames' index range with the pixels taken from the path indices
    (render_sample_path's tiled branch, src/render.d/gi.c:88-95): the same paths, whoever renders them"""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_tile_worker, args=(world, port, d), nprocs=world, join=True)
        reduced = np.load(os.path.join(d, "reduced_tiles.npy"))
    scene = make_scene(SCENE_0010, width=96, height=64, max_verts=4)
    with oracle_pixels():
        single, cnt, _ = oracle_render(scene, 0, 6 * scene.width * scene.height, threads=1)
        # the pixel of path i is (i mod W H): one sample per pixel per frame
        from helpers import oracle_records
        rec = oracle_records(scene, 5 * scene.width * scene.height - 3, 6)
    assert np.array_equal(np.floor(rec["pixel_i"]), np.float32([93, 94, 95, 0, 1, 2])) and np.array_equal(np.floor(rec["pixel_j"]), np.float32([63, 63, 63, 0, 0, 0]))
    assert np.allclose(reduced, single, rtol=1e-5, atol=1e-4)
    assert reduced.sum() > 0
    # three tile owners, one of them without a tile in the last row: still every pixel exactly once
    with oracle_pixels():
        three = sum(oracle_render_tiles(scene, 0, 6, g, 3, threads=2)[0] for g in range(3))
    assert np.allclose(three, single, rtol=1e-5, atol=1e-4)
