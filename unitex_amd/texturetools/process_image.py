"""Reference-image preprocessing (host side, PIL): crop to the matte's bounding box, rescale to `scale` of the frame,
centre on a background colour -- TextureTools/texturetools/image/process_image.py:10-74 (`get_bbox`, `preprocess`).

The matte itself comes from RMBG-2.0 in the reference ([3p] model + checkpoint, not available here): this build takes
the alpha channel of the input image when it has one (the function's own RGBA branch), otherwise treats the whole
frame as foreground."""
import numpy as np
from PIL import Image, ImageOps


def get_bbox(mask: np.ndarray):
    assert mask.ndim == 2
    rows = np.where(mask.sum(-1) > 0)[0]
    cols = np.where(mask.sum(-2) > 0)[0]
    return np.array([cols.min(), rows.min(), cols.max(), rows.max()])


def preprocess(image: Image.Image, alpha=None, H=2048, W=2048, scale=0.8, color="white", return_alpha=False):
    image = ImageOps.exif_transpose(image)
    rgb = image.convert("RGB")
    if alpha is None:
        if image.mode == "RGBA" and np.sum(np.array(image.getchannel("A")) > 0) < image.size[0] * image.size[1] - 8:
            alpha = image.getchannel("A")
        else:
            alpha = Image.new("L", image.size, 255)          # no matting model: everything is foreground
    box = get_bbox(np.array(alpha))
    x1, y1, x2, y2 = box
    dy, dx = y2 - y1, x2 - x1
    s = min(H * scale / dy, W * scale / dx)
    Ht, Wt = int(dy * s), int(dx * s)
    ox, oy = int((W - Wt) / 2), int((H - Ht) / 2)
    target = np.array([ox, oy, ox + Wt, oy + Ht])
    rgbc = rgb.crop(box).resize((Wt, Ht))
    alphac = alpha.crop(box).resize((Wt, Ht))
    alphat = Image.new("L", (W, H))
    alphat.paste(alphac, target)
    out = Image.new("RGBA", (W, H), color)
    out.paste(rgbc, target, alphac)
    out.putalpha(alphat)
    return (out, alpha) if return_alpha else out
