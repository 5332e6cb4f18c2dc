"""Image readers the reference notebooks import from ``crf.utils`` (caller side of the hot path;
mirrors crf/utils.py:46-51 read_image, :59-91 read_pfm, :93-109 read_pgm).  Own implementation;
the reference's read_pgm carries two NameErrors (`filename`, `byteorder`), fixed here."""
import re

import numpy as np

try:                                    # the reference module leaks these to ``from crf.utils import *`` users
    from PIL import Image  # noqa: F401
    import matplotlib.pyplot as plt  # noqa: F401
except Exception:                       # headless / missing backends: readers below still work
    pass


def read_image(imgname):
    """RGB image as float64 [h, w, 3] in [0, 1]."""
    from PIL import Image

    return np.asarray(Image.open(imgname).convert("RGB")).astype(float) / 255.0


def read_pfm(file):
    """Middlebury .pfm disparity (grey `Pf` or colour `PF`, first channel plane layout as the
    reference returns it): float array [h, w], row 0 at the top."""
    with open(file, "rb") as f:
        kind = f.readline().decode("latin-1").strip()
        if kind not in ("PF", "Pf"):
            raise ValueError(f"Not a PFM file: {file!r}")
        channels = 3 if kind == "PF" else 1
        width, height = (int(t) for t in re.findall(r"\d+", f.readline().decode("latin-1")))
        scale = float(f.readline().decode("latin-1").strip())
        data = np.frombuffer(f.read(width * height * channels * 4), dtype=("<f4" if scale < 0 else ">f4"))
    return np.flip(data[:width * height].reshape(height, width).astype(np.float64), axis=0)


def read_pgm(file):
    """Raw (P5) PGM as an integer array [h, w]."""
    with open(file, "rb") as f:
        buf = f.read()
    m = re.match(rb"P5\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s+(?:#[^\n]*\n\s*)*(\d+)\s", buf)
    if m is None:
        raise ValueError("Not a raw PGM file: '%s'" % file)
    width, height, maxval = (int(g) for g in m.groups())
    dtype = "u1" if maxval < 256 else ">u2"
    return np.frombuffer(buf, dtype=dtype, count=width * height, offset=m.end()).reshape(height, width)
