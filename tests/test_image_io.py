"""Image readers/writers of the host library (the reference uses stb + tinyexr, src/util/imageio.cpp:419-726; both are
absent here, so the formats are read directly: PNG and EXR ZIP/ZIPS through zlib).  Test files are produced with
Python's own zlib / struct so the readers are checked against an independent encoder."""
import struct
import zlib

import numpy as np
import pytest

from luisarender_amd.scene import HostError, load_image, save_image

RNG = np.random.default_rng(3)


def _png(path, arr, color_type, depth, filters=(0, 1, 2, 3, 4), palette=None):
    h, w = arr.shape[:2]
    raw = bytearray()
    data = arr.astype(">u2" if depth == 16 else np.uint8).reshape(h, -1)
    rows = [r.tobytes() for r in data]
    bpp = max(1, len(rows[0]) // w)
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)]
        out = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (x - p) & 255
        raw += bytes([ft]) + out
        prev = row
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    z = zlib.compress(bytes(raw), 6)
    body = chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    body += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + body)


@pytest.mark.parametrize("color_type,samples,depth", [(0, 1, 8), (2, 3, 8), (6, 4, 8), (4, 2, 8), (2, 3, 16), (6, 4, 16)])
def test_png_reader(tmp_path, color_type, samples, depth):
    w, h = 37, 21
    arr = RNG.integers(0, 1 << depth, (h, w, samples))
    path = str(tmp_path / "t.png")
    _png(path, arr, color_type, depth)
    img, channels = load_image(path)
    scale = float((1 << depth) - 1)
    assert img.shape == (h, w, 4) and channels == samples
    expect = np.ones((h, w, 4), np.float32)
    if samples <= 2:
        expect[..., :3] = arr[..., :1] / scale
        if samples == 2:
            expect[..., 3] = arr[..., 1] / scale
    else:
        expect[..., :samples] = arr / scale
    assert np.allclose(img, expect, atol=1e-6)


def test_png_palette(tmp_path):
    w, h = 16, 9
    palette = RNG.integers(0, 256, (7, 3))
    idx = RNG.integers(0, 7, (h, w, 1))
    path = str(tmp_path / "p.png")
    _png(path, idx, 3, 8, palette=palette)
    img, channels = load_image(path)
    assert channels == 3 and np.allclose(img[..., :3], palette[idx[..., 0]] / 255.0, atol=1e-6) and (img[..., 3] == 1).all()


def _rle(data):
    """OpenEXR's run-length code: count byte n >= 0 -> next byte repeated n + 1 times (runs of 3..128); n < 0 -> -n literals"""
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        run = 1
        while i + run < n and data[i + run] == data[i] and run < 128:
            run += 1
        if run >= 3:
            out += bytes([run - 1, data[i]])
            i += run
        else:
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
                j += 1
            out += bytes([(256 - (j - i)) & 255]) + data[i:j]
            i = j
    return bytes(out)


def _exr(path, img, compression, half):
    """scanline OpenEXR writer: compression 0 (none) / 2 (ZIPS) / 3 (ZIP), channels B G R (alphabetical) as half or float"""
    h, w = img.shape[:2]
    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    ptype = 1 if half else 2
    chlist = b"".join(n + b"\0" + struct.pack("<IBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([compression])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if compression in (3, 5) else 1
    chunks = []
    for y0 in range(0, h, lines):
        raw = b""
        for y in range(y0, min(y0 + lines, h)):
            for c in (2, 1, 0):  # B, G, R
                raw += img[y, :, c].astype(np.float16 if half else np.float32).tobytes()
        if compression == 5:
            # PXR24 (ImfPxr24Compressor): per row and channel the byte PLANES of the left-neighbour differences, most significant
            # first -- 2 planes of a half, the top 3 bytes of a float (its low 8 mantissa bits are dropped) -- deflated as one stream
            planes = bytearray()
            for y in range(y0, min(y0 + lines, h)):
                for c in (2, 1, 0):
                    if half:
                        v = img[y, :, c].astype(np.float16).view(np.uint16).astype(np.int64)
                        d = np.diff(v, prepend=0) & 0xFFFF
                        planes += bytes((d >> 8).astype(np.uint8)) + bytes((d & 255).astype(np.uint8))
                    else:
                        v = (img[y, :, c].astype(np.float32).view(np.uint32) >> 8).astype(np.int64)  # (the test image is exact in 24 bits)
                        d = np.diff(v, prepend=0) & 0xFFFFFF
                        planes += bytes((d >> 16).astype(np.uint8)) + bytes(((d >> 8) & 255).astype(np.uint8)) + bytes((d & 255).astype(np.uint8))
            payload = zlib.compress(bytes(planes))
        elif compression:
            n = len(raw)
            t = bytearray(n)
            t[: (n + 1) // 2] = raw[0::2]
            t[(n + 1) // 2:] = raw[1::2]
            d = bytearray(n)
            d[0] = t[0]
            for i in range(1, n):
                d[i] = (t[i] - t[i - 1] + 128 + 256) & 255
            z = _rle(bytes(d)) if compression == 1 else zlib.compress(bytes(d))
            payload = z if len(z) < n else raw
        else:
            payload = raw
        chunks.append(struct.pack("<iI", y0, len(payload)) + payload)
    table_at = len(head)
    offsets, pos = [], table_at + 8 * len(chunks)
    for c in chunks:
        offsets.append(pos)
        pos += len(c)
    open(path, "wb").write(head + b"".join(struct.pack("<Q", o) for o in offsets) + b"".join(chunks))


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("half", [True, False])
def test_exr_reader_with_zip_compression(tmp_path, compression, half):
    w, h = 53, 37  # not a multiple of the 16-line ZIP block
    img = (RNG.random((h, w, 3)) * 8).astype(np.float16 if half else np.float32).astype(np.float32)
    img[5:9] = 0.25  # compressible runs
    path = str(tmp_path / "t.exr")
    _exr(path, img, compression, half)
    got, channels = load_image(path)
    assert channels == 3 and got.shape == (h, w, 4)
    assert np.array_equal(got[..., :3], img) and (got[..., 3] == 1).all()


@pytest.mark.parametrize("half", [True, False])
def test_exr_reader_with_pxr24_compression(tmp_path, half):
    """EXR compression 5: a float keeps its top 24 bits, so the picture is made of values that have no more"""
    w, h = 53, 37
    img = (RNG.random((h, w, 3)) * 8).astype(np.float32)
    img[5:9] = 0.25
    img[20, 10:14] = [[-3.5, 1e-3, 1e4]] * 4  # sign changes and big steps: the differences wrap around
    img = img.astype(np.float16) if half else (img.view(np.uint32) & np.uint32(0xFFFFFF00)).view(np.float32)
    img = img.astype(np.float32)
    path = str(tmp_path / "t.exr")
    _exr(path, img, 5, half)
    got, channels = load_image(path)
    assert channels == 3 and got.shape == (h, w, 4)
    assert np.array_equal(got[..., :3], img) and (got[..., 3] == 1).all()
    data = bytearray(open(path, "rb").read())
    data[-20] ^= 0x55  # a damaged deflate stream is an error, not garbage pixels
    open(path, "wb").write(bytes(data))
    with pytest.raises(Exception, match="PXR24|Corrupt|Truncated"):
        load_image(path)


def test_exr_and_hdr_round_trip_of_our_own_writer(tmp_path):
    img = np.concatenate([RNG.random((19, 23, 3)).astype(np.float32) * 4, np.ones((19, 23, 1), np.float32)], -1)
    save_image(str(tmp_path / "a.exr"), img)
    got, _ = load_image(str(tmp_path / "a.exr"))
    assert np.allclose(got[..., :3], img[..., :3], rtol=1e-3, atol=1e-4)  # the writer stores halfs or floats
    save_image(str(tmp_path / "a.hdr"), img)
    got, _ = load_image(str(tmp_path / "a.hdr"))
    step = img[..., :3].max(axis=-1, keepdims=True) / 128.0  # RGBE: 8-bit mantissas under the exponent of the largest channel
    assert (np.abs(got[..., :3] - img[..., :3]) <= step).all()


def _test_picture(w, h, seed=3):
    """smooth colour ramps + a few hard edges + noise: exercises DC / AC coefficients and chroma upsampling"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(x / 7.0), 127 + 100 * np.cos(y / 5.0 + x / 11.0), (x * 255 // max(w - 1, 1))], axis=-1)
    img[h // 3: h // 2, w // 4: w // 2] = (250, 20, 30)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("mode,subsampling,progressive,size,restart", [
    ("RGB", 0, False, (64, 48), 0),     # 4:4:4 baseline
    ("RGB", 1, False, (67, 45), 0),     # 4:2:2, ragged size
    ("RGB", 2, False, (70, 53), 0),     # 4:2:0 (the usual texture file)
    ("RGB", 2, True, (70, 53), 0),      # progressive: spectral selection + successive approximation
    ("RGB", 0, True, (33, 17), 0),
    ("L", 0, False, (40, 40), 0),       # greyscale
    ("L", 0, True, (41, 23), 0),
    ("RGB", 2, False, (120, 90), 4),    # restart intervals
    ("RGB", 2, False, (1, 1), 0),
    ("RGB", 2, False, (9, 3), 0),
])
def test_jpeg_reader_against_libjpeg(tmp_path, mode, subsampling, progressive, size, restart):
    """Our decoder restates stb_image's numerics (integer LLM IDCT, 3:1 triangle chroma upsampling, fixed-point YCbCr);
    PIL decodes with libjpeg's (ISLOW IDCT, its own fancy upsampling): the same picture within a few code values."""
    w, h = size
    pic = _test_picture(w, h)
    im = PIL.fromarray(pic if mode == "RGB" else pic[..., 1], mode)
    path = tmp_path / "t.jpg"
    kw = dict(quality=92, progressive=progressive)
    if mode == "RGB":
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    im.save(path, **kw)
    got, ch = load_image(str(path))
    ref = np.asarray(PIL.open(path).convert("RGB"), np.float32) / 255.0
    assert ch == (3 if mode == "RGB" else 1) and got.shape == (h, w, 4) and (got[..., 3] == 1).all()
    err = np.abs(got[..., :3] - ref)
    # IDCT / upsampling / colour rounding differ by a code value or two; 4:2:0 edges a little more
    assert err.mean() < (1.2 if subsampling else 0.6) / 255 and np.quantile(err, 0.99) <= 6 / 255 and err.max() <= 24 / 255, (err.mean() * 255, err.max() * 255)
    # and both are the picture that went in, up to the quantisation
    assert np.abs(got[..., :3] - (pic if mode == "RGB" else pic[..., 1:2].repeat(3, -1)) / 255.0).mean() < 8 / 255


def test_jpeg_flat_blocks_are_exact(tmp_path):
    """a constant picture has only DC coefficients: every decoder must reproduce the same flat value"""
    im = PIL.fromarray(np.full((16, 24, 3), (200, 100, 50), np.uint8), "RGB")
    im.save(tmp_path / "flat.jpg", quality=100, subsampling=0)
    got, _ = load_image(str(tmp_path / "flat.jpg"))
    ref = np.asarray(PIL.open(tmp_path / "flat.jpg").convert("RGB"), np.float32) / 255.0
    assert np.abs(got[..., :3] - ref).max() <= 1 / 255 + 1e-6
    assert np.ptp(got[..., 0]) == 0 and np.ptp(got[..., 1]) == 0


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "P"])
def test_bmp_reader(tmp_path, mode):
    pic = _test_picture(37, 21)
    if mode == "RGBA":
        src = np.concatenate([pic, (pic[..., :1] // 2 + 100)], axis=-1)
    elif mode == "L":
        src = pic[..., 0]
    else:
        src = pic
    im = PIL.fromarray(src, "RGB" if mode == "P" else mode)
    if mode == "P":
        im = im.quantize(64)
    im.save(tmp_path / "t.bmp")
    got, ch = load_image(str(tmp_path / "t.bmp"))
    ref = np.asarray(PIL.open(tmp_path / "t.bmp").convert("RGBA"), np.float32) / 255.0
    if mode == "RGBA":  # PIL drops the alpha byte of a 32-bit BI_RGB file when reading; stb_image (and this reader) keep it
        ref[..., 3] = src[..., 3] / 255.0
    assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-6), mode
    assert ch == (4 if mode == "RGBA" else 3)


@pytest.mark.parametrize("mode,rle", [("RGB", False), ("RGB", True), ("RGBA", False), ("RGBA", True), ("L", False), ("L", True), ("P", False)])
def test_tga_reader(tmp_path, mode, rle):
    pic = _test_picture(29, 18)
    pic[4:9, 3:20] = (10, 200, 30)  # runs for the RLE packets
    src = np.concatenate([pic, (pic[..., :1] // 2 + 100)], axis=-1) if mode == "RGBA" else (pic[..., 0] if mode == "L" else pic)
    im = PIL.fromarray(src, "RGB" if mode == "P" else mode)
    if mode == "P":
        im = im.quantize(32)
    im.save(tmp_path / "t.tga", compression="tga_rle" if rle else None)
    got, ch = load_image(str(tmp_path / "t.tga"))
    ref = np.asarray(PIL.open(tmp_path / "t.tga").convert("RGBA"), np.float32) / 255.0
    assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-6), (mode, rle)
    assert ch == {"RGB": 3, "RGBA": 4, "L": 1, "P": 3}[mode]


def test_unsupported_formats_fail_loudly(tmp_path):
    p = tmp_path / "x.jpg"
    p.write_bytes(b"\xff\xd8\xff")
    with pytest.raises(HostError):
        load_image(str(p))
    q = tmp_path / "x.gif"
    q.write_bytes(b"GIF89a")
    with pytest.raises(HostError):
        load_image(str(q))


def test_corrupt_files_fail_with_an_error_not_a_crash():
    """regression corpus of an AddressSanitizer fuzzing pass over the readers (mutated / truncated files that used to read or
    write out of bounds): every one must come back as a HostError or as a finite image"""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fuzz", "*")))
    assert len(files) >= 7
    for f in files:
        try:
            img, _ = load_image(f)
            assert np.isfinite(img).all(), f
        except HostError:
            pass


def test_exr_header_cannot_make_the_reader_allocate_gigabytes(tmp_path):
    """A few-hundred-byte file whose dataWindow claims 60000 x 60000 pixels (54 GB of float4) is refused before anything is allocated."""
    import struct
    good = tmp_path / "good.exr"
    save_image(str(good), np.full((4, 6, 4), 0.5, np.float32))
    raw = bytearray(good.read_bytes())
    at = raw.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    assert struct.unpack_from("<4i", raw, at) == (0, 0, 5, 3)
    struct.pack_into("<4i", raw, at, 0, 0, 59999, 59999)
    bad = tmp_path / "bad.exr"
    bad.write_bytes(bytes(raw))
    with pytest.raises(HostError, match="too large"):
        load_image(str(bad))
    # tall but narrow: the pixel count is small, the offset table the header promises is longer than the file
    struct.pack_into("<4i", raw, at, 0, 0, 5, 59999)
    bad.write_bytes(bytes(raw))
    with pytest.raises(HostError, match="too large"):
        load_image(str(bad))
    # under the 2^28-pixel cap and with a long-enough offset table, but 16 bytes per pixel are more than the file could ever expand to
    # (an 8 KB file with a 16384 x 16384 window used to get its 4 GiB): 8 KB of padding behind the chunks, 16384 x 512
    padded = bytes(raw) + b"\0" * 8192
    raw2 = bytearray(padded)
    struct.pack_into("<4i", raw2, at, 0, 0, 16383, 511)
    bad.write_bytes(bytes(raw2))
    with pytest.raises(HostError, match="too large"):
        load_image(str(bad))
    got, _ = load_image(str(good))
    assert got.shape == (4, 6, 4) and np.allclose(got[..., :3], 0.5)


# ---------------------------------------------------------------------------------------------------------------- EXR PIZ
@pytest.mark.parametrize("case", ["half_smooth", "half_noise", "float_rgba", "constant", "odd_sizes", "many_values", "tall"])
def test_exr_piz_reader_against_an_independent_encoder(tmp_path, case):
    """csrc/host/image_io.cpp namespace piz (bitmap LUT, inverse wavelet, canonical Huffman + run-length symbol) reads what
    tests/exr_piz_encoder.py -- written separately from the same published format -- produces, exactly.  (No PIZ file and no
    other EXR codec exist on this machine: the reader is NOT pinned against a real OpenEXR file; DESIGN.md says so.)"""
    from exr_piz_encoder import write_exr_piz
    rng = np.random.default_rng(5)
    half, channels = True, "RGB"
    if case == "half_smooth":  # few distinct values: the 14-bit wavelet, long runs
        y, x = np.mgrid[0:40, 0:52]
        img = np.stack([x / 52, y / 40, (x + y) / 92], -1).astype(np.float32)
        img = np.round(img * 16) / 16
    elif case == "half_noise":  # > 2^14 distinct half values: the 16-bit wavelet
        img = rng.normal(0, 30, (70, 300, 3)).astype(np.float32)
    elif case == "float_rgba":
        img, half, channels = rng.random((33, 17, 4)).astype(np.float32) * 10 - 2, False, "RGBA"
    elif case == "constant":  # one value: bitmap of one bit, a two-symbol code
        img = np.full((5, 9, 3), 0.25, np.float32)
    elif case == "odd_sizes":
        img = rng.random((37, 13, 3)).astype(np.float32)
    elif case == "many_values":
        img, half = (rng.random((64, 64, 3)) * 1000).astype(np.float32), False
    else:  # several 32-line chunks, the last one short, width 1
        img = rng.random((71, 1, 3)).astype(np.float32)
    path = tmp_path / "piz.exr"
    coded = write_exr_piz(str(path), img, half=half, channels=channels)
    assert coded == (img.shape[0] + 31) // 32  # every chunk went through the coder
    assert path.read_bytes().find(b"compression\0compression\0\x01\0\0\0\x04") > 0
    got, ch = load_image(str(path))
    want = img.astype(np.float16).astype(np.float32) if half else img
    assert got.shape == (img.shape[0], img.shape[1], 4) and ch == len(channels)
    assert np.array_equal(got[..., :len(channels)], want)
    if len(channels) == 3:
        assert (got[..., 3] == 1).all()


def test_exr_piz_corrupt_chunks_fail_with_an_error(tmp_path):
    from exr_piz_encoder import write_exr_piz
    img = np.random.default_rng(1).random((40, 24, 3)).astype(np.float32)
    path = tmp_path / "piz.exr"
    assert write_exr_piz(str(path), img) == 2
    raw = bytearray(path.read_bytes())
    rng = np.random.default_rng(2)
    failures = 0
    for trial in range(300):
        bad = bytearray(raw)
        at = int(rng.integers(len(raw) // 4, len(raw)))
        bad[at] ^= 1 << int(rng.integers(8))
        (tmp_path / "bad.exr").write_bytes(bytes(bad))
        try:
            got, _ = load_image(str(tmp_path / "bad.exr"))
            assert got.shape == (40, 24, 4)
        except HostError:
            failures += 1
    assert failures > 10  # symbol and bit counts are checked exactly; a flip that keeps both decodes to other pixels, never crashes
    (tmp_path / "bad.exr").write_bytes(bytes(raw[:len(raw) - 20]))
    with pytest.raises(HostError):
        load_image(str(tmp_path / "bad.exr"))


def test_the_device_decodes_an_8_bit_code_to_the_float_the_host_readers_make():
    """dev_shade.h: texel_at keeps 8-bit images as 8-bit texels (lrhip_set_texture_storage) and must hand the shading code the very floats
    the host readers made: b * (1 / 255.f) (the PNG reader) is its first product; b / 255.f (JPEG / BMP / TGA, image_codecs.cpp) is that
    product plus ONE Newton step, fma(fma(-q, 255, b), 1 / 255.f, q).  Restated here in exact rational arithmetic with one rounding per
    operation (what fp32 multiply / fma do), for all 256 codes, against numpy's correctly rounded fp32 division.  (lrhip_upload_scene runs
    the same check on the device's own function before it packs anything; the GPU side is
    tests/test_gpu_parity.py::test_byte_texels_decode_to_the_floats_the_host_made.)"""
    from fractions import Fraction

    def f32(x: Fraction) -> Fraction:  # round to nearest even at 24 bits (normal range: every value here is in [2^-30, 256])
        if x == 0:
            return x
        sign, x = (-1, -x) if x < 0 else (1, x)
        e = x.numerator.bit_length() - x.denominator.bit_length()
        if Fraction(2) ** e > x:
            e -= 1
        ulp = Fraction(2) ** (e - 23)
        n, rest = divmod(x, ulp)
        if rest * 2 > ulp or (rest * 2 == ulp and n % 2 == 1):
            n += 1
        return sign * n * ulp

    k = f32(Fraction(1, 255))
    assert float(k) == float(np.float32(1.0) / np.float32(255.0))
    for b in range(256):
        q = f32(b * k)
        assert float(q) == float(np.float32(b) * np.float32(float(k)))           # form 1: the PNG reader's product
        r = f32(b - q * 255)                                                     # fma(-q, 255, b): one rounding
        v = f32(r * k + q)                                                       # fma(r, 1 / 255.f, q)
        assert float(v) == float(np.float32(b) / np.float32(255.0)), b           # form 2: the division, correctly rounded
        assert float(f32(r * 0 + q)) == float(q)                                 # the same expression with the correction's weight 0 is form 1
    assert sum(1 for b in range(256) if f32(b * k) != f32(Fraction(b, 255))) > 20  # (the two forms do differ: the step is needed)
