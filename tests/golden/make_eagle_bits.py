#!/usr/bin/env python3
"""Derives the reference's payload bits from its input image and stores them as a fixture.

`file_reader.m:4-11` reads `eagle.tiff` (identical in all five task directories), binarises it with
`imbinarize` (Otsu, `graythresh` on the 256-bin histogram) and takes the first `Size_Buffer` elements
in column-major order.  The image is a DATA file of the reference, not code; the fixture holds the
129 600 bits packed (16 KB) so that the published PAPR numbers of `Task 2/README.md:54,:70-71`
(which depend on the image payload) can be checked where /root/reference is absent.

Run in the build container:  python tests/golden/make_eagle_bits.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/Task 2/eagle.tiff"


def graythresh_u8(img):
    """Otsu's method as MATLAB's graythresh does it for uint8: maximise the between-class variance over the
    256-bin histogram; ties -> mean of the maximisers; level normalised to [0, 1]."""
    counts = np.bincount(img.ravel(), minlength=256).astype(np.float64)
    p = counts / counts.sum()
    omega = np.cumsum(p)
    mu = np.cumsum(p * np.arange(1, 257))
    mu_t = mu[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        sigma_b = (mu_t * omega - mu) ** 2 / (omega * (1 - omega))
    sigma_b[~np.isfinite(sigma_b)] = -np.inf
    mx = sigma_b.max()
    idx = np.mean(np.flatnonzero(sigma_b == mx)) + 1
    return (idx - 1) / 255.0


def main():
    img = np.asarray(Image.open(SRC))
    assert img.shape == (360, 360) and img.dtype == np.uint8, (img.shape, img.dtype)
    level = graythresh_u8(img)
    bw = img > level * 255                       # imbinarize: I > T for the class range
    bits = bw.ravel(order="F").astype(np.uint8)  # file_reader.m:11 linear (column-major) indexing
    np.savez_compressed(os.path.join(HERE, "eagle_bits.npz"), packed=np.packbits(bits), n_bits=bits.size,
                        otsu_level_255=round(level * 255))
    print("level*255 =", level * 255, "ones =", bits.mean(), "first 64:", "".join(map(str, bits[:64])))


if __name__ == "__main__":
    main()
