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


@pytest.mark.parametrize("world,width,height,partition", [(2, 48, 32, "xor"), (3, 37, 23, "xor"), (2, 40, 30, "seed")])
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
