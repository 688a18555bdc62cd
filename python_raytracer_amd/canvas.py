"""The tile consumer of the trace path: Window.draw_tile (reference init.py:185-190).

The reference blits every finished tile onto a persistent SRCALPHA canvas; the tile's alpha,
round(min(1, energy + shutter) * 255) (init.py:141), makes that blit the motion blur.  `Canvas` keeps the canvas on
the device and blends tiles into it with vrt_canvas_blit, so a frame never leaves HBM between the march and the
display hand-off.  The blend restates pygame 2's ALPHA_BLEND; pygame cannot be run in the build environment, so this
module is parity-unpinned.  Post-FX (spill, iris, bloom, scaling: init.py:207-253) are not built.
"""
import ctypes as C

from . import _native as nat


class Canvas:
    def __init__(self, width, height, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("python_raytracer_amd needs a ROCm GPU: there is no CPU fallback")
        self._torch = torch
        self.width, self.height = int(width), int(height)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        # pg.Surface(window, pg.SRCALPHA): transparent black
        self.rgba8 = torch.zeros((self.height, self.width, 4), dtype=torch.uint8, device=self.device)

    def blit(self, tile, pixels=None):
        """canvas.blit(tile, (0, 0)) (init.py:188).  tile: a RenderResult (its own pixel list is used) or an
        [H, W, 4] uint8 tensor / RGBA bytes of the full window as Camera.tile returns them."""
        torch = self._torch
        px = None
        if hasattr(tile, "image_u8"):
            px = tile.pixels if pixels is None else pixels
            tile = tile.image_u8
        elif isinstance(tile, (bytes, bytearray)):
            import numpy as np
            tile = torch.from_numpy(np.frombuffer(tile, np.uint8).reshape(self.height, self.width, 4).copy()).to(self.device)
        if tuple(tile.shape) != (self.height, self.width, 4) or tile.dtype != torch.uint8:
            raise ValueError("tile must be a [%d, %d, 4] uint8 image" % (self.height, self.width))
        d_px, n = None, 0
        if px is not None:
            d_px = px.tensor if hasattr(px, "tensor") else torch.as_tensor(px, dtype=torch.int32, device=self.device).contiguous()
            n = int(d_px.shape[0])
        with torch.cuda.device(self.device):
            nat.check(nat.lib().vrt_canvas_blit(self.rgba8.data_ptr(), tile.contiguous().data_ptr(), self.width, self.height,
                                                d_px.data_ptr() if d_px is not None else None, n,
                                                torch.cuda.current_stream().cuda_stream), "vrt_canvas_blit")
        return self.rgba8

    def tobytes(self):
        """pg.image.tobytes(canvas, "RGBA")"""
        return self.rgba8.cpu().numpy().tobytes()
