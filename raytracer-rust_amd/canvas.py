"""`Canvas` (src/graphics.rs:12-81): the framebuffer a render returns.

The device writes both forms of every pixel: the linear colour narrowed to FP32
(`Canvas::pixels`, row-major x + y*width, src/graphics.rs:40-46) and the RGBA8 image the
reference derives from it with `(c * 255.0) as u8` (src/graphics.rs:59-74).  This class
only holds and serialises them.
"""
import numpy as np


class Canvas:
    def __init__(self, width, height, rgb=None, rgba8=None, stats=None):
        self.width, self.height = int(width), int(height)
        self.rgb = rgb if rgb is not None else np.zeros((self.height, self.width, 3), dtype=np.float32)
        self.rgba8 = rgba8
        self.stats = stats

    def get_pixel(self, x, y):
        return tuple(float(v) for v in self.rgb[y, x])

    def set_pixel(self, x, y, color):
        self.rgb[y, x] = color
        self.rgba8 = None

    def as_slice(self):
        return self.rgb.reshape(-1, 3)

    def to_image(self):
        """RGBA8 image [height, width, 4]; uses the device's cast when present, otherwise the
        same truncating, saturating cast on the host."""
        if self.rgba8 is not None:
            return self.rgba8
        v = self.rgb.astype(np.float64) * 255.0
        v = np.where(np.isnan(v), 0.0, v)
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        out[..., :3] = np.clip(np.trunc(v), 0, 255).astype(np.uint8)
        out[..., 3] = 255
        return out

    def save_to_png(self, path):
        """`Canvas::save_to_png` (src/graphics.rs:77-81): RGBA8, non-interlaced PNG."""
        from PIL import Image

        Image.fromarray(self.to_image(), mode="RGBA").save(path, format="PNG")
