"""N>1 path on CPU: world_size-2 (and 3) gloo process groups run the rank partition + tile gather of
python_raytracer_amd.multigpu, with the CPU oracle standing in for the per-rank renderer.  The assembled image on
rank 0 must equal the single-process frame bit for bit (the multi-GPU = single-GPU identity of SURVEY.md 8e)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, partition, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from python_raytracer_amd.multigpu import rank_pixels, rank_pixel_counts, TileGather
    from python_raytracer_amd.data import pixel_partition
    sc = ol.default_scene()
    st = ol.make_settings(width=width, height=height, samples=2, max_bounces=4)
    px = rank_pixels(width, height, world, rank, partition, 2)
    if partition == "xor":  # the shard IS the reference's settings.pixels[rank] for threads == world
        assert np.array_equal(px, pixel_partition(width, height, world)[rank].array)
    assert rank_pixel_counts(width, height, world, partition, 2)[rank] == len(px)
    o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, px, want_rays=False, want_traversed=False)
    local = torch.from_numpy(o["pix_mean"].astype(np.float32))
    g = TileGather(width, height, 4, torch.float32, torch.device("cpu"), partition=partition, samples=2)
    img = None
    for _ in range(2):                       # buffers are reusable across frames
        img = g(local)
    # pipelined form, one frame in flight: frame k is collected after frame k + 1 was produced; the returned image
    # is a view of the gather's own buffer, valid until the next collect()
    base = img.clone() if rank == 0 else None
    g.submit(local * 2)
    nxt = local * 4
    img2 = g.collect()
    assert rank != 0 or torch.equal(img2, base * 2)
    win = torch.zeros((height, width, 4))   # the rank's window image: own pixels painted, the rest 0
    win[torch.from_numpy(px[:, 1].astype(np.int64)), torch.from_numpy(px[:, 0].astype(np.int64))] = nxt
    g.submit(window=win)
    img4 = g.collect()
    if rank == 0:
        assert torch.equal(img4, base * 4)
        img = base
    # whole-job ray count the way bench.py aggregates it
    t = torch.tensor([float(o["n_rays"])], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        q.put((img.numpy().copy(), float(t[0])))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,partition", [(2, 48, 32, "xor"), (3, 37, 23, "xor"), (2, 40, 30, "seed"), (3, 45, 29, "tiles")])
def test_gloo_gather_equals_single_process(world, width, height, partition):
    sys.path.insert(0, HERE)
    import oracle_lib as ol
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, width, height, partition, q)) for r in range(world)]
    for p in procs:
        p.start()
    img, nrays = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = ol.default_scene()
    st = ol.make_settings(width=width, height=height, samples=2, max_bounces=4)
    full = np.concatenate(ol.pixel_lists(width, height, 1))
    o = ol.render(sc, st, sc.cam_pos, sc.cam_rot, sc.cam_lens, full, want_rays=False, want_traversed=False)
    ref = np.zeros((height, width, 4), np.float32)
    ref[full[:, 1], full[:, 0]] = o["pix_mean"].astype(np.float32)
    assert np.array_equal(img, ref)
    assert nrays == o["n_rays"]


# ------------------------------------------------------------------------------------------------ culling feedback
def _trav_box(pos, dist_max, cs):
    """Camera._trav_box for an unrotated camera (velocity bound 1): origin (world coords) and dims of the key box."""
    import math
    r = int(math.ceil((float(dist_max) + 1.0 + cs / 2.0) / cs)) + 1
    o = [(int(math.floor(v / cs)) - r) * cs for v in pos]
    return np.array(o, np.int64), 2 * r + 1


def _culling_worker(rank, world, port, q):
    """Replays tests/golden/culling_sequence.npz (five frames of the real reference with culling on, one thread) with
    the pixels sharded over `world` ranks: per frame select chunks from the UNION of last frame's traversed keys
    (multigpu.union_traversed), render the shard (CPU oracle), gather the tiles."""
    import json
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from python_raytracer_amd.multigpu import rank_pixels, TileGather, union_traversed
    z = np.load(os.path.join(ol.GOLDEN, "culling_sequence.npz"))
    sc = ol.default_scene()
    s = json.loads(bytes(z["settings"]).decode())
    st = ol.make_settings(**{k: s[k] for k in ol.DEFAULT_SETTINGS if k in s})
    w, h, cs = st["width"], st["height"], 16
    px = rank_pixels(w, h, world, rank, "xor", st["samples"])
    g = TileGather(w, h, 4, torch.float32, torch.device("cpu"), partition="xor", samples=st["samples"])
    prev = np.zeros((0, 3))
    grew = False
    ok = True
    for it in range(5):
        pos = z["pos_%d" % it]
        pres, res = ol.select_chunks(sc.origin, sc.dims, cs, sc.present, pos, float(z["dist_max"][0]),
                                     int(z["chunk_lod"][0]), True, prev)
        ok = ok and np.array_equal(pres, z["present_%d" % it]) and np.array_equal(res, z["res_%d" % it])
        grid = ol.Scene.camera_grid(sc.grid_lod0, sc.origin, sc.dims, cs, pres, res)
        cs_ = ol.Scene(sc.origin, sc.dims, cs, pres, res, grid, sc.materials)
        o = ol.render(cs_, st, pos, z["cam_rot"], z["cam_lens"][0], px)
        img = g(torch.from_numpy(o["pix_mean"].astype(np.float32)))
        if rank == 0:
            ok = ok and np.array_equal(img.numpy(), z["pix_%d" % it].astype(np.float32))
        # this rank's visit keys over the camera's box, in the format vrt_render_tile writes (UINT64_MAX = never)
        origin, n = _trav_box(pos, st["dist_max"], cs)
        keys = torch.full((n * n * n,), -1, dtype=torch.int64)
        own = o["traversed"]
        c = (own - origin) // cs
        assert ((c >= 0) & (c < n)).all()
        keys[torch.from_numpy((c[:, 0] * n + c[:, 1]) * n + c[:, 2])] = torch.arange(len(own), dtype=torch.int64) << 12
        union_traversed(keys)
        idx = torch.nonzero(keys != -1).flatten().numpy()
        union = np.stack([idx // (n * n), (idx // n) % n, idx % n], 1) * cs + origin
        # the union over the ranks is the single-process traversed set of the fixture (order aside)
        exp = z["traversed_%d" % it].astype(np.int64)
        ok = ok and sorted(map(tuple, union.tolist())) == sorted(map(tuple, exp.tolist()))
        grew = grew or len(union) > len(own)
        prev = union.astype(np.float64)
    q.put((rank, bool(ok), bool(grew)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_culling_sequence_world2_equals_single_process():
    """With culling on (the reference default) an N-rank run must cull against the union of every rank's traversed
    chunks (reference init.py:189, 393): world size 2 reproduces the single-process chunk tables, LODs and images of
    the reference's five-frame culling sequence, and a rank's own list alone is a strict subset."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_culling_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [True, True], got
    assert any(g[2] for g in got), "the shards' traversed sets never differed: the test would not notice a missing union"


def test_merge_traversed_is_unsigned_min():
    from python_raytracer_amd.multigpu import merge_traversed
    a = torch.tensor([-1, 5 << 12, -1, 7], dtype=torch.int64)
    b = torch.tensor([-1, 3 << 12, 9, -1], dtype=torch.int64)
    assert merge_traversed([a, b]).tolist() == [-1, 3 << 12, 9, 7]
    assert merge_traversed([a]).tolist() == a.tolist()
