"""World building (Goxel import -> Sprite -> Object -> dense world grid, python_raytracer_amd/world.py) against
tests/golden/world_build.npz, produced by running the real reference's Sprite.load / Object / Window.chunk_update
(data.py:253-427, 430-494, 589-600; init.py:398-444) on the synthetic assets in tests/golden/assets."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from python_raytracer_amd import Material, make_settings
from python_raytracer_amd.lib import vec3, rgb, material
from python_raytracer_amd.world import Sprite, Object, build_world

ASSETS = os.path.join(ol.GOLDEN, "assets")


def build_from_fixture(z):
    mats = [Material(function=material, albedo=rgb(*[int(v) for v in row[:3]]), roughness=row[3], absorption=row[4],
                     ior=row[5], energy=row[6]) for row in z["materials"]]
    cmap = {str(c): m for c, m in zip(z["colours"], mats)}
    s = json.loads(bytes(z["settings"]).decode())
    st = make_settings(**{k: s[k] for k in ("width", "height", "samples", "max_bounces", "dist_max", "chunk_lod")})
    st.culling = False
    cam_pos = vec3(*[float(v) for v in z["cam_pos"]])
    objs = []
    for fn, size, lod, pos, rot in zip(z["spec_files"], z["spec_size"], z["spec_lod"], z["spec_pos"], z["spec_rot"]):
        spr = Sprite(size=vec3(*[float(v) if v % 1 else int(v) for v in size]), frames=1, lod=int(lod))
        spr.load([os.path.join(ASSETS, str(fn))], cmap)
        ob = Object(pos=vec3(*[float(v) if v % 1 else int(v) for v in pos]), rot=vec3(*[int(v) for v in rot]), sprite=spr)
        ob.update(cam_pos, st)
        objs.append(ob)
    return mats, st, objs


def test_world_build_matches_reference():
    z = np.load(os.path.join(ol.GOLDEN, "world_build.npz"))
    mats, st, objs = build_from_fixture(z)
    assert np.array_equal([[o.sprite.size.x, o.sprite.size.y, o.sprite.size.z] for o in objs], z["sprite_size"])
    assert np.array_equal([[o.mins.x, o.mins.y, o.mins.z] for o in objs], z["obj_mins"])
    assert np.array_equal([[o.maxs.x, o.maxs.y, o.maxs.z] for o in objs], z["obj_maxs"])
    assert np.array_equal([o.visible for o in objs], z["obj_visible"])
    w = build_world(objs, 16)
    # material numbering here is first-seen; the fixture's is the order of z["materials"]
    remap = np.zeros(len(w.materials) + 1, np.uint8)
    for k, m in enumerate(w.materials):
        remap[k + 1] = 1 + mats.index(m)
    # same chunk-aligned box up to empty border chunks
    lo = np.minimum(w.origin, z["origin"])
    hi = np.maximum(w.origin + w.dims * 16, z["origin"] + z["dims"] * 16)
    a = np.zeros(tuple(hi - lo), np.uint8)
    b = np.zeros(tuple(hi - lo), np.uint8)
    o1, o2 = w.origin - lo, z["origin"] - lo
    a[o1[0]:o1[0] + w.grid.shape[0], o1[1]:o1[1] + w.grid.shape[1], o1[2]:o1[2] + w.grid.shape[2]] = remap[w.grid]
    g = z["grid_lod0"]
    b[o2[0]:o2[0] + g.shape[0], o2[1]:o2[1] + g.shape[1], o2[2]:o2[2] + g.shape[2]] = g
    assert np.array_equal(a, b)
    assert int(w.present.sum()) == int(z["present"].sum())
    # single-position API (Sprite.get_voxel with rotation, reference data.py:417-419) agrees with the vectorised build
    # for an object nothing else overlaps
    ob = objs[2]
    for x in range(12):
        for y in range(0, 12, 3):
            for z in range(12):
                m = ob.sprite.get_voxel(None, vec3(x, y, z), ob.rot)
                got = w.grid[tuple(np.array([int(ob.mins.x) + x, int(ob.mins.y) + y, int(ob.mins.z) + z]) - w.origin)]
                assert got == (0 if m is None else 1 + w.materials.index(m))


REF_VOXELS = "/root/reference/mods/default/voxels"


@pytest.mark.skipif(not os.path.isdir(REF_VOXELS), reason="the reference's asset files only exist in the build container")
def test_default_mod_assets_build_the_fixture_scene():
    """The reference's own Goxel exports (castle 209 k voxels, material cube, player) through Sprite.load / Object /
    build_world, placed as mods/default/init.py:175-214 places them, give exactly the voxel grid the real reference
    produced for the scene fixture (tests/golden/scene_default.npz) -- including the mirrored-X off-by-one that drops
    the castle's x = 0 column.  Reads the asset files in place; nothing of them is copied into the repository."""
    sc = ol.default_scene()
    names = list(sc.names)
    mat = {n: Material(function=material, albedo=rgb(*[int(v) for v in row[:3]]), roughness=row[3], absorption=row[4],
                       ior=row[5], energy=row[6]) for n, row in zip(names, sc.materials)}
    cube = lambda special: {"7f7f7f": mat["material"], "ffffff": mat[special]}  # noqa: E731
    spec = [  # (file, sprite size, colour -> material, object position): mods/default/init.py:175-214
        ("castle.txt.gz", (128, 64, 128), {"000000": mat["metal"], "3f3f3f": mat["stone_dark"], "7f7f7f": mat["stone_gray"],
                                            "bfbfbf": mat["stone_light"], "ffffff": mat["stone_marble"]}, (0, 0, 0)),
        ("material.txt.gz", (12, 12, 12), cube("material_rough"), (-56, -16, 56)),
        ("material.txt.gz", (12, 12, 12), cube("material_light"), (12, -24, 24)),
        ("material.txt.gz", (12, 12, 12), cube("material_scatter"), (48, -24, -48)),
        ("material.txt.gz", (12, 12, 12), cube("material_glass"), (-4, 18, 16)),
        ("material.txt.gz", (12, 12, 12), cube("material_shiny"), (-56, 18, 16)),
        ("material.txt.gz", (12, 12, 12), cube("material_mist"), (-36, 18, -36)),
        ("player.txt.gz", (12, 16, 12), {"7f7f7f": mat["player"]}, (-12, 0, -8)),
    ]
    st = make_settings(dist_max=192)
    cam_pos = vec3(*[float(v) for v in sc.cam_pos])
    objs = []
    for fn, size, cmap, pos in spec:
        spr = Sprite(size=vec3(*size), frames=1, lod=0)
        spr.load([os.path.join(REF_VOXELS, fn)], cmap)
        ob = Object(pos=vec3(*pos), rot=vec3(0, 0, 0), sprite=spr)
        assert ob.update(cam_pos, st)
        objs.append(ob)
    w = build_world(objs, 16)
    remap = np.zeros(len(w.materials) + 1, np.uint8)
    for k, m in enumerate(w.materials):
        remap[k + 1] = 1 + names.index([n for n in names if mat[n] is m][0])
    lo = np.minimum(w.origin, sc.origin)
    hi = np.maximum(w.origin + w.dims * 16, sc.origin + sc.dims * 16)
    a = np.zeros(tuple(hi - lo), np.uint8)
    b = np.zeros(tuple(hi - lo), np.uint8)
    o1, o2 = w.origin - lo, sc.origin - lo
    a[o1[0]:o1[0] + w.grid.shape[0], o1[1]:o1[1] + w.grid.shape[1], o1[2]:o1[2] + w.grid.shape[2]] = remap[w.grid]
    g = sc.grid_lod0
    b[o2[0]:o2[0] + g.shape[0], o2[1]:o2[1] + g.shape[1], o2[2]:o2[2] + g.shape[2]] = g
    assert int((b != 0).sum()) > 200000
    assert np.array_equal(a, b)


def test_loader_errors_like_reference(tmp_path):
    p = tmp_path / "bad.txt"
    p.write_text("1 2 3 ff0000\n4 4\n")
    spr = Sprite(size=vec3(4, 4, 4))
    with pytest.raises(ValueError):
        spr.load([str(p)], {"ff0000": Material(function=material, albedo=rgb(1, 2, 3), roughness=0, absorption=1, ior=1,
                                               energy=0)})
    with pytest.raises(ValueError):
        spr.load([str(tmp_path / "model.vox")], {})
    assert Sprite(size=vec3(7, 5, 9)).size.tuple() == (8, 6, 10)
    assert Sprite(size=vec3(6.5, 4, 4)).size.tuple() == (6, 4, 4)


@pytest.mark.gpu
def test_world_build_renders_like_reference():
    """Assets -> build_world -> Camera.set_world_scene -> chunk_update (LOD selection) -> tile: image and traversed
    list equal to what the real reference rendered for the same objects."""
    from python_raytracer_amd import Camera
    from python_raytracer_amd.lib import quaternion
    z = np.load(os.path.join(ol.GOLDEN, "world_build.npz"))
    mats, st, objs = build_from_fixture(z)
    cam = Camera(settings=st)
    cam.pos = vec3(*[float(v) for v in z["cam_pos"]])
    cam.rot = quaternion(0.0, 0.0, 0.0, 1.0)
    cam.lens = float(z["cam_lens"][0])
    w = build_world(objs, 16)
    cam.set_world_scene(w.packed())
    table = cam.chunk_update(None).cpu().numpy().view(np.uint32).reshape(tuple(w.dims))
    # LOD map: compare on the overlap with the fixture's chunk box
    off = (z["origin"] - w.origin) // 16
    sub = table[off[0]:off[0] + z["dims"][0], off[1]:off[1] + z["dims"][1], off[2]:off[2] + z["dims"][2]]
    assert np.array_equal((sub != 0).astype(np.uint8), z["cam_present"])
    assert np.array_equal((sub >> 24).astype(np.uint8), z["cam_res"])
    r = cam.render(0)
    px = r.pixels
    assert np.array_equal(r.rgba_f32.cpu().numpy(), z["pix"][px[:, 1], px[:, 0]].astype(np.float32))
    assert np.array_equal(np.array(r.traversed(16)).reshape(-1, 3), z["traversed"])


def _device_grid(dw, ps):
    """Dense [X, Y, Z] grid and presence map out of a DeviceWorld's packed blocks."""
    from python_raytracer_amd.scene import unpack_blocks
    cs = dw.chunk_size
    dims = [int(v) for v in dw.dims]
    table = ps.device_tensors["chunk_table"].cpu().numpy().view(np.uint32).reshape(dims)
    blocks = unpack_blocks(ps.device_tensors["voxels"].cpu().numpy().reshape(-1, cs ** 3), cs)
    grid = blocks.reshape(dims[0], dims[1], dims[2], cs, cs, cs).transpose(0, 3, 1, 4, 2, 5).reshape(
        dims[0] * cs, dims[1] * cs, dims[2] * cs)
    return table, grid


def _assert_same_world(dw, ps, w):
    table, grid = _device_grid(dw, ps)
    assert np.array_equal(dw.origin, w.origin) and np.array_equal(dw.dims, w.dims)
    assert [id(m) for m in dw.materials] == [id(m) for m in w.materials]
    assert np.array_equal(grid, w.grid)
    assert np.array_equal((table != 0).astype(np.uint8), w.present)
    n = np.arange(table.size, dtype=np.uint32).reshape(table.shape) + 1
    assert np.array_equal(table[table != 0], (n | (1 << 24))[table != 0])


@pytest.mark.gpu
def test_device_world_equals_host_build_and_reference_render():
    """vrt_voxelize (DeviceWorld.build) gives the voxel blocks, chunk presence and materials of build_world() -- which
    is pinned to the real reference by test_world_build_matches_reference -- also after objects moved and turned, and
    the frame rendered from the device-built world is the reference's."""
    from python_raytracer_amd import Camera
    from python_raytracer_amd.lib import quaternion
    from python_raytracer_amd.world import DeviceWorld
    z = np.load(os.path.join(ol.GOLDEN, "world_build.npz"))
    mats, st, objs = build_from_fixture(z)
    dw = DeviceWorld(16)
    ps = dw.build(objs)
    w = build_world(objs, 16)
    _assert_same_world(dw, ps, w)
    cam = Camera(settings=st)
    cam.pos = vec3(*[float(v) for v in z["cam_pos"]])
    cam.rot = quaternion(0.0, 0.0, 0.0, 1.0)
    cam.lens = float(z["cam_lens"][0])
    cam.set_world_scene(ps)
    cam.chunk_update(None)
    r = cam.render(0)
    px = r.pixels
    assert np.array_equal(r.rgba_f32.cpu().numpy(), z["pix"][px[:, 1], px[:, 0]].astype(np.float32))
    assert np.array_equal(np.array(r.traversed(16)).reshape(-1, 3), z["traversed"])
    # move and turn the objects: same models, new boxes (the world box itself changes size)
    for k, ob in enumerate(objs):
        ob.pos = ob.pos + vec3(3 * k - 4, 2.5 * (k % 2), -7 + 5 * k)
        ob.rot = vec3(90 * (k % 4), 90 * ((k + 1) % 4), 90 * ((k + 2) % 4))
        ob.set_sprite(ob.sprite)
    ps = dw.build(objs)
    _assert_same_world(dw, ps, build_world(objs, 16))
    # nothing visible
    for ob in objs:
        ob.visible = False
    ps = dw.build(objs)
    assert int(ps.device_tensors["chunk_table"].abs().sum()) == 0 and list(dw.dims) == [1, 1, 1]


@pytest.mark.gpu
def test_device_world_random_objects():
    """Random overlapping objects with every quarter-turn combination, cubic and non-cubic models (a rotation about an
    axis is ignored unless the two other extents are equal, data.py:344-371), fractional positions and LOD sprites."""
    from python_raytracer_amd.world import DeviceWorld
    rng = np.random.RandomState(11)
    mats = [Material(function=material, albedo=rgb(10 * i, 20, 30), roughness=0.1, absorption=1, ior=0, energy=0)
            for i in range(1, 7)]
    for cs in (8, 16):
        objs = []
        for k in range(9):
            size = [int(rng.choice([4, 6, 8, 10])) for _ in range(3)]
            if k % 3 == 0:
                size[1] = size[2] = size[0]
            lod = int(rng.choice([0, 0, 1]))
            spr = Sprite(size=vec3(*size), frames=1, lod=lod)
            vox = {}
            for _ in range(60):
                p = tuple(int(rng.randint(0, s)) for s in size)
                vox[p] = mats[rng.randint(len(mats))]
            spr.get_frame(0).set_voxels(vox, True)
            pos = [float(rng.randint(-20, 20)) + float(rng.choice([0, 0, 0.5])) for _ in range(3)]
            ob = Object(pos=vec3(*pos), rot=vec3(*[int(rng.choice([0, 90, 180, 270, 360, -90])) for _ in range(3)]),
                        sprite=spr)
            ob.visible = bool(k != 4)
            objs.append(ob)
        dw = DeviceWorld(cs)
        _assert_same_world(dw, dw.build(objs), build_world(objs, cs))


def _merge_world(dw, cs):
    """build_world() over the DeviceWorld's own merge order and chunk box (the incremental update keeps its box)."""
    w = build_world(list(dw._order), cs)
    origin, dims = np.asarray(dw.origin, np.int64), np.asarray(dw.dims, np.int64)
    grid = np.zeros(tuple(dims * cs), np.uint8)
    if w.grid.any():
        a = np.asarray(w.origin, np.int64) - origin
        grid[a[0]:a[0] + w.grid.shape[0], a[1]:a[1] + w.grid.shape[1], a[2]:a[2] + w.grid.shape[2]] = w.grid
    # build_world numbers materials in first-use order of ITS object order; map them onto the device's ids
    remap = np.zeros(256, np.uint8)
    for k, m in enumerate(w.materials):
        remap[k + 1] = 1 + [id(x) for x in dw.materials].index(id(m))
    grid = remap[grid]
    present = grid.reshape(dims[0], cs, dims[1], cs, dims[2], cs).any(axis=(1, 3, 5)).astype(np.uint8)
    return grid, present


@pytest.mark.gpu
def test_device_world_incremental_update_moves_one_object_in_a_large_world():
    """DeviceWorld.update (vrt_voxelize with a chunk list): in a 24 x 4 x 24-chunk world one object moves, one turns,
    one vanishes and one appears; only the chunks their old and new boxes touch are rebuilt, the result equals a
    rebuild of everything in the reference's merge order (a re-voxelised object moves to the end: init.py:403, 427),
    and the untouched chunks' bytes are not written."""
    import torch
    from python_raytracer_amd.world import DeviceWorld
    rng = np.random.RandomState(5)
    cs = 16
    mats = [Material(function=material, albedo=rgb(20 * i, 10, 40), roughness=0.2, absorption=1, ior=0, energy=0)
            for i in range(1, 9)]

    def sprite(size):
        spr = Sprite(size=vec3(*size), frames=1, lod=0)
        vox = {}
        for _ in range(size[0] * size[1] * size[2] // 3):
            vox[tuple(int(rng.randint(0, s)) for s in size)] = mats[rng.randint(len(mats))]
        spr.get_frame(0).set_voxels(vox, True)
        return spr

    objs = []
    for k in range(40):
        size = [8, 8, 8] if k % 2 else [int(rng.choice([6, 10, 12])) for _ in range(3)]
        pos = [float(rng.randint(-180, 180)), float(rng.randint(-24, 24)), float(rng.randint(-180, 180))]
        ob = Object(pos=vec3(*pos), rot=vec3(0, 90 * (k % 4), 0), sprite=sprite(size))
        ob.visible = True
        objs.append(ob)
    # two objects that overlap: the later one in the merge order wins where both have a voxel
    objs[1].move(vec3(objs[0].pos.x + 3, objs[0].pos.y, objs[0].pos.z - 2))
    dw = DeviceWorld(cs)
    ps = dw.build(objs)
    n_chunks = int(np.prod(np.asarray(dw.dims, np.int64)))
    assert n_chunks > 1000
    table, grid = _device_grid(dw, ps)
    g0, p0 = _merge_world(dw, cs)
    assert np.array_equal(grid, g0) and np.array_equal((table != 0).astype(np.uint8), p0)
    # nothing changed: nothing rebuilt
    ps, rebuilt = dw.update(objs)
    assert rebuilt == 0
    # one object moves (inside the box), the overlapped one turns, one vanishes, one appears
    before = ps.device_tensors["voxels"].clone()
    objs[0].move(vec3(objs[0].pos.x + 21, objs[0].pos.y - 5, objs[0].pos.z + 40))
    objs[1].rot = vec3(0, objs[1].rot.y + 90, 0)
    objs[7].visible = False
    new = Object(pos=vec3(objs[20].pos.x + 2, objs[20].pos.y + 1, objs[20].pos.z), rot=vec3(0, 0, 0), sprite=sprite([8, 8, 8]))
    new.visible = True
    objs.insert(3, new)
    ps, rebuilt = dw.update(objs)
    assert 0 < rebuilt <= 60 and rebuilt < n_chunks // 20
    assert [id(o) for o in dw._order[-3:]] == [id(objs[0]), id(objs[1]), id(new)]   # re-voxelised objects go last
    table, grid = _device_grid(dw, ps)
    g1, p1 = _merge_world(dw, cs)
    assert np.array_equal(grid, g1) and np.array_equal((table != 0).astype(np.uint8), p1)
    # ... the same voxels a rebuild of every chunk in that merge order gives
    full = DeviceWorld(cs)
    fs = full.build(list(dw._order))
    if list(full.dims) == list(dw.dims):   # (material numbers differ: first use in each world's own history)
        lut = np.zeros(256, np.int64)
        lut[1:1 + len(full.materials)] = [1 + [id(m) for m in dw.materials].index(id(m)) for m in full.materials]
        assert np.array_equal(lut[_device_grid(full, fs)[1]], grid)
    changed = (before != ps.device_tensors["voxels"]).view(-1, cs ** 3).any(1).sum().item()
    assert 0 < changed <= rebuilt
    # an object leaves the box: full rebuild with a new box
    objs[5].move(vec3(400.0, 0.0, 0.0))
    ps, rebuilt = dw.update(objs)
    assert rebuilt == int(np.prod(np.asarray(dw.dims, np.int64)))
    table, grid = _device_grid(dw, ps)
    g2, p2 = _merge_world(dw, cs)
    assert np.array_equal(grid, g2) and np.array_equal((table != 0).astype(np.uint8), p2)


# ---------------------------------------------------------------------------------------- merge order after re-draws
def _same_voxels(origin_a, grid_a, origin_b, grid_b):
    """Two dense grids over (possibly different) chunk-aligned boxes hold the same voxels."""
    origin_a, origin_b = np.asarray(origin_a, np.int64), np.asarray(origin_b, np.int64)
    lo = np.minimum(origin_a, origin_b)
    hi = np.maximum(origin_a + grid_a.shape, origin_b + grid_b.shape)
    a, b = np.zeros(tuple(hi - lo), np.uint8), np.zeros(tuple(hi - lo), np.uint8)
    o = origin_a - lo
    a[o[0]:o[0] + grid_a.shape[0], o[1]:o[1] + grid_a.shape[1], o[2]:o[2] + grid_a.shape[2]] = grid_a
    o = origin_b - lo
    b[o[0]:o[0] + grid_b.shape[0], o[1]:o[1] + grid_b.shape[1], o[2]:o[2] + grid_b.shape[2]] = grid_b
    return np.array_equal(a, b)


def _redraw_sequence(objs, st, cam_pos, seq):
    """The re-draws tests/golden/make_golden.py drove through the reference's Window.chunk_update (world_update.npz):
    yields (tag, ids of the objects that redraw) after applying each tick's changes to `objs`."""
    yield "1", {id(objs[0])}                                  # the slab redraws alone
    objs[1].move(vec3(*[float(v) if v % 1 else int(v) for v in seq["cube_pos_2"]]))
    objs[1].update(cam_pos, st)
    yield "2", {id(objs[0]), id(objs[1])}                     # both redraw, the cube moved by one voxel


def test_merge_order_after_redraws_matches_reference():
    """Which object wins an overlap after objects redraw separately and then together (reference init.py:398-429: a
    redrawn object is deleted from the chunks_objects dict and re-inserted as the loop over data.objects meets it).
    DeviceWorld.merge_order against the reference's own dict order and voxels (tests/golden/world_update.npz)."""
    from python_raytracer_amd.world import DeviceWorld
    z = np.load(os.path.join(ol.GOLDEN, "world_build.npz"))
    seq = np.load(os.path.join(ol.GOLDEN, "world_update.npz"))
    mats, st, objs = build_from_fixture(z)
    cam_pos = vec3(*[float(v) for v in z["cam_pos"]])
    vis = [o for o in objs if o.visible]
    order = list(vis)
    assert [objs.index(o) for o in order] == seq["order_0"].tolist()
    for tag, changed in _redraw_sequence(objs, st, cam_pos, seq):
        stay, moved = DeviceWorld.merge_order(order, vis, changed)
        order = stay + moved
        assert [objs.index(o) for o in order] == seq["order_" + tag].tolist()
        w = build_world(order, 16)
        remap = np.zeros(len(w.materials) + 1, np.uint8)
        for k, m in enumerate(w.materials):
            remap[k + 1] = 1 + mats.index(m)
        assert _same_voxels(w.origin, remap[w.grid], seq["origin_" + tag], seq["grid_" + tag])
    # the two ticks really differ where the slab and the cube overlap (else the fixture would pin nothing)
    assert not _same_voxels(seq["origin_0"], seq["grid_0"], seq["origin_1"], seq["grid_1"])


@pytest.mark.gpu
def test_device_world_update_keeps_the_reference_merge_order():
    """DeviceWorld.update through the same re-draws: the voxels on the device are the reference's after every tick."""
    from python_raytracer_amd.world import DeviceWorld
    z = np.load(os.path.join(ol.GOLDEN, "world_build.npz"))
    seq = np.load(os.path.join(ol.GOLDEN, "world_update.npz"))
    mats, st, objs = build_from_fixture(z)
    cam_pos = vec3(*[float(v) for v in z["cam_pos"]])
    dw = DeviceWorld(16)
    ps = dw.build(objs)

    def check(tag):
        _, grid = _device_grid(dw, ps)
        remap = np.zeros(len(dw.materials) + 1, np.uint8)
        for k, m in enumerate(dw.materials):
            remap[k + 1] = 1 + mats.index(m)
        assert _same_voxels(dw.origin, remap[grid], seq["origin_" + tag], seq["grid_" + tag]), tag

    check("0")
    for tag, changed in _redraw_sequence(objs, st, cam_pos, seq):
        for o in objs:
            if id(o) in changed:
                o.redraw = True
        ps, rebuilt = dw.update(objs)
        assert rebuilt > 0
        assert [objs.index(o) for o in dw._order] == seq["order_" + tag].tolist()
        check(tag)
