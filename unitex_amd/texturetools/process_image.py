"""Reference-image preprocessing (host side, PIL): the matte's bounding box is cut out, scaled so that its longer
relative side fills `scale` of the output frame, and centred on a flat background.  Replaces `preprocess` of
TextureTools/texturetools/image/process_image.py:31-74 (output bytes pinned by fixture G10).

Matte source, in order: an explicit `alpha` image; the input's own alpha channel when it is a real matte (more than 8
transparent pixels -- the reference's test, process_image.py:45); otherwise the whole frame.  The reference's third source,
the RMBG-2.0 network ([3p] model + checkpoint), does not exist in this build: pass `matting_fn(image) -> RGBA` to plug one in.
"""
from typing import Callable, NamedTuple, Optional

import numpy as np
from PIL import Image, ImageOps


class Box(NamedTuple):
    """pixel rectangle in PIL order; right / bottom follow whoever produced the box (see foreground_box)."""
    left: int
    top: int
    right: int
    bottom: int

    @property
    def width(self):
        return self.right - self.left

    @property
    def height(self):
        return self.bottom - self.top


def foreground_box(matte: np.ndarray) -> Box:
    """tight box of the non-zero matte pixels, max INDEX as right / bottom (so the box is one pixel short of PIL's
    exclusive convention -- the reference crops with exactly this box, process_image.py:10-18,55,63)."""
    if matte.ndim != 2:
        raise ValueError("matte must be a single-channel image")
    occupied_cols = np.flatnonzero(matte.any(axis=0))
    occupied_rows = np.flatnonzero(matte.any(axis=1))
    if occupied_cols.size == 0:
        raise ValueError("matte is empty")
    return Box(int(occupied_cols[0]), int(occupied_rows[0]), int(occupied_cols[-1]), int(occupied_rows[-1]))


def select_matte(image: Image.Image, alpha: Optional[Image.Image], matting_fn: Optional[Callable]) -> Image.Image:
    if alpha is not None:
        return alpha
    if image.mode == "RGBA":
        own = image.getchannel("A")
        opaque = int(np.count_nonzero(np.asarray(own)))
        if opaque < image.width * image.height - 8:
            return own
    if matting_fn is not None:
        return matting_fn(image).getchannel("A")
    return Image.new("L", image.size, 255)


def fit_centered(src: Box, frame_w: int, frame_h: int, scale: float) -> Box:
    """destination rectangle: uniform zoom that brings the box to `scale` of the frame along its tighter axis, truncated to
    whole pixels, centred with truncation (same float expressions as the reference so the sizes agree to the pixel)."""
    zoom = min(frame_h * scale / src.height, frame_w * scale / src.width)
    out_w, out_h = int(src.width * zoom), int(src.height * zoom)
    left, top = int((frame_w - out_w) / 2), int((frame_h - out_h) / 2)
    return Box(left, top, left + out_w, top + out_h)


def preprocess(image: Image.Image, alpha: Optional[Image.Image] = None, H=2048, W=2048, scale=0.8, color="white",
               return_alpha=False, matting_fn: Optional[Callable] = None):
    """-> RGBA frame W x H: the foreground composited over `color` through its (resampled) matte, alpha = the placed matte."""
    image = ImageOps.exif_transpose(image)
    matte = select_matte(image, alpha, matting_fn)
    src = foreground_box(np.asarray(matte))
    dst = fit_centered(src, W, H, scale)
    size = (dst.width, dst.height)
    fg = image.convert("RGB").crop(tuple(src)).resize(size)
    fg_matte = matte.crop(tuple(src)).resize(size)
    frame = Image.new("RGBA", (W, H), color)
    frame.paste(fg, tuple(dst), fg_matte)
    placed = Image.new("L", (W, H))
    placed.paste(fg_matte, tuple(dst))
    frame.putalpha(placed)
    return (frame, matte) if return_alpha else frame
