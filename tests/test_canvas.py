"""Window.draw_tile's alpha-over blit (reference init.py:185-190) -- the first step of SURVEY.md 8f row N2.

PARITY UNPINNED: pygame is not installable in the build environment and the reference ships no image of a blended
canvas, so the blend restates pygame 2's published ALPHA_BLEND macro (oracle/vrt_oracle.c: orc_canvas_blit) and these
tests only pin the GPU kernel to that restatement and check the properties the reference relies on."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol


def orc_blit(canvas, tile):
    L = ol.lib()
    L.orc_canvas_blit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.orc_canvas_blit.restype = None
    out = np.ascontiguousarray(canvas, np.uint8).copy()
    t = np.ascontiguousarray(tile, np.uint8)
    L.orc_canvas_blit(out.ctypes.data, t.ctypes.data, out.size // 4)
    return out


def test_restated_blend_properties():
    rng = np.random.default_rng(1)
    canvas = rng.integers(0, 256, (40, 30, 4), dtype=np.uint8)
    tile = rng.integers(0, 256, (40, 30, 4), dtype=np.uint8)
    # a transparent source pixel leaves the canvas as it is (why blitting only a tile's own pixels is the same blit)
    t0 = tile.copy()
    t0[..., 3] = 0
    t0[..., :3] = 0
    keep = canvas[..., 3] != 0
    assert np.array_equal(orc_blit(canvas, t0)[keep], canvas[keep])
    # onto a transparent canvas the tile is copied
    assert np.array_equal(orc_blit(np.zeros_like(canvas), tile), tile)
    # an opaque source replaces the colour (within the macro's >> 8 rounding) and keeps the canvas opaque
    c1 = canvas.copy()
    c1[..., 3] = 255
    t1 = tile.copy()
    t1[..., 3] = 255
    o = orc_blit(c1, t1)
    assert np.abs(o[..., :3].astype(int) - t1[..., :3].astype(int)).max() <= 1 and (o[..., 3] == 255).all()
    # repeated blits of the same tile converge towards it: the reference's motion blur
    c = c1.copy()
    th = tile.copy()
    th[..., 3] = 128
    d0 = np.abs(c[..., :3].astype(int) - th[..., :3].astype(int)).mean()
    for _ in range(12):
        c = orc_blit(c, th)
    assert np.abs(c[..., :3].astype(int) - th[..., :3].astype(int)).mean() < 0.05 * d0 + 1.5


@pytest.mark.gpu
def test_canvas_blit_kernel_equals_restatement_and_tiles_compose():
    """vrt_canvas_blit against orc_canvas_blit on random images, with and without a pixel list, and a frame composed
    from the 4 thread tiles of Camera.render equals the blit of the whole frame."""
    import torch
    from python_raytracer_amd import Canvas
    from gpu_util import camera_for, settings_store
    rng = np.random.default_rng(2)
    W, H = 96, 54
    cv = Canvas(W, H)
    ref = np.zeros((H, W, 4), np.uint8)
    for it in range(4):
        tile = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        if it % 2:
            tile[rng.random((H, W)) < 0.5] = 0                    # transparent holes
        cv.blit(torch.from_numpy(tile).cuda())
        ref = orc_blit(ref, tile)
        assert np.array_equal(cv.rgba8.cpu().numpy(), ref)
    # tiles of a render: 4 threads, each blits only its own pixels
    sc = ol.default_scene()
    st = ol.make_settings(width=W, height=H, samples=2, max_bounces=4, threads=4)
    cam = camera_for(sc, settings_store(st), sc.cam_pos, sc.cam_rot, sc.cam_lens)
    a, b = Canvas(W, H), Canvas(W, H)
    for frame in range(3):                                       # successive frames accumulate (motion blur)
        full = np.zeros((H, W, 4), np.uint8)
        for t in range(4):
            r = cam.render(t, want_f32=False, want_traversed=False)
            a.blit(r)                                             # RenderResult: own pixel list
            b.blit(r.image_u8)                                    # whole-window blit of the same tile
            full |= r.image_u8.cpu().numpy()
        assert torch.equal(a.rgba8, b.rgba8)
        ref = orc_blit(ref if frame else np.zeros_like(full), full) if frame else full.copy()
        assert np.array_equal(a.rgba8.cpu().numpy(), ref)
    assert a.tobytes() == ref.tobytes()
    # Camera.tile's bytes are accepted as well
    image, _, _ = cam.tile(0)
    c = Canvas(W, H)
    c.blit(image)
    assert np.array_equal(c.rgba8.cpu().numpy(), np.frombuffer(image, np.uint8).reshape(H, W, 4))
