"""Image files in and out: the role cv::imread / cv::imwrite play in the reference's main() (Source.cpp:623,635).
Decoding and encoding are host work and stay in a library (Pillow); the blur runs on the GPU through the C ABI."""
import numpy as np


def imread(path):
    """uint8 array [rows, cols, 3] (channel order as stored: the blur treats the channels independently)"""
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))


def imwrite(path, image):
    from PIL import Image
    a = np.ascontiguousarray(image, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected a uint8 image [rows, cols, 3]")
    Image.fromarray(a, "RGB").save(path)


def blur_file(src_path, dst_path, sigma, ctx=None, **kw):
    """imread -> pffft_(image, sigma) on the GPU -> imwrite: main() of the reference for flag 3 (Source.cpp:611-641)"""
    from .api import BlurContext
    own = ctx is None
    c = BlurContext(0) if own else ctx
    try:
        out = c.pffft_(imread(src_path), sigma, **kw)
    finally:
        if own:
            c.close()
    imwrite(dst_path, out)
    return out
