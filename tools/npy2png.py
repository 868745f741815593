#!/usr/bin/env python3
"""Tone-map a float RGBA .npy (or .exr written by this repo) to an 8-bit PNG for eyeballing."""
import struct
import sys
import zlib

import numpy as np


def write_png(path, img8):
    h, w, _ = img8.shape
    raw = b"".join(b"\x00" + img8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    png += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    open(path, "wb").write(png)


def tonemap(rgb):
    rgb = np.clip(rgb, 0, None)
    srgb = np.where(rgb <= 0.0031308, rgb * 12.92, 1.055 * np.power(rgb, 1 / 2.4) - 0.055)
    return (np.clip(srgb, 0, 1) * 255 + 0.5).astype(np.uint8)


if __name__ == "__main__":
    a = np.load(sys.argv[1])
    write_png(sys.argv[2], tonemap(a[..., :3]))
