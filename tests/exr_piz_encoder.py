"""A small OpenEXR PIZ ENCODER (test infrastructure): writes single-part scanline EXR files with compression 4 so that the PIZ
decoder of csrc/host/image_io.cpp has something to read -- there is no PIZ file and no other EXR codec on this machine.

Written from the published format (OpenEXR's ImfPizCompressor / ImfHuf / ImfWav), independently of the C++ decoder: forward
value-compaction LUT from the occupancy bitmap, forward 2-D wavelet (wenc14 below 2^14 distinct values, wenc16 otherwise), a
Huffman code over the 16-bit words plus one run-length symbol, code lengths packed as 6-bit fields with zero runs, canonical
codes with the longest codes holding the smallest values.
"""
import heapq
import struct

import numpy as np

SHORT_ZEROCODE_RUN, LONG_ZEROCODE_RUN = 59, 63
SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN
LONGEST_LONG_RUN = 255 + SHORTEST_LONG_RUN


class BitWriter:
    def __init__(self):
        self.out, self.c, self.lc, self.bits = bytearray(), 0, 0, 0

    def put(self, n, value):
        self.c = (self.c << n) | (value & ((1 << n) - 1))
        self.lc += n
        self.bits += n
        while self.lc >= 8:
            self.lc -= 8
            self.out.append((self.c >> self.lc) & 0xff)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc > 0:
            self.out.append((self.c << (8 - self.lc)) & 0xff)
            self.c, self.lc = 0, 0
        return bytes(self.out)


def _wenc14(a, b):
    a, b = int(a), int(b)
    a, b = (a - 65536 if a >= 32768 else a), (b - 65536 if b >= 32768 else b)  # as signed shorts
    m, d = (a + b) >> 1, a - b
    return m & 0xffff, d & 0xffff


def _wenc16(a, b):
    ao = (int(a) + 0x8000) & 0xffff
    m, d = (ao + int(b)) >> 1, ao - int(b)
    if d < 0:
        m = (m + 0x8000) & 0xffff
    return m & 0xffff, d & 0xffff


def wavelet_encode(plane, w14):
    """plane: int array [ny, nx] of 16-bit words, transformed in place (finest level first)"""
    enc = _wenc14 if w14 else _wenc16
    ny, nx = plane.shape
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        y = 0
        while y <= ny - p2:
            x = 0
            while x <= nx - p2:
                i00, i01 = enc(plane[y, x], plane[y, x + p])
                i10, i11 = enc(plane[y + p, x], plane[y + p, x + p])
                plane[y, x], plane[y + p, x] = enc(i00, i10)
                plane[y, x + p], plane[y + p, x + p] = enc(i01, i11)
                x += p2
            if nx & p:
                plane[y, x], plane[y + p, x] = enc(plane[y, x], plane[y + p, x])
            y += p2
        if ny & p:
            x = 0
            while x <= nx - p2:
                plane[y, x], plane[y, x + p] = enc(plane[y, x], plane[y, x + p])
                x += p2
        p, p2 = p2, p2 << 1


def huffman_compress(words):
    """words: list of ints in [0, 65535] -> the hufCompress byte string"""
    freq = {}
    for s in words:
        freq[s] = freq.get(s, 0) + 1
    im, iM = min(freq), max(freq) + 1
    freq[iM] = 1  # the run-length symbol
    # code lengths by the textbook algorithm
    heap = [(f, i, (s,)) for i, (s, f) in enumerate(sorted(freq.items()))]
    heapq.heapify(heap)
    length = {s: 0 for s in freq}
    tick = len(heap)
    if len(heap) == 1:
        length[heap[0][2][0]] = 1
    while len(heap) > 1:
        fa, _, sa = heapq.heappop(heap)
        fb, _, sb = heapq.heappop(heap)
        for s in sa + sb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, tick, sa + sb))
        tick += 1
    assert max(length.values()) <= 58
    # canonical codes: from the longest length down
    count = [0] * 59
    for l in length.values():
        count[l] += 1
    first, c = [0] * 59, 0
    for l in range(58, 0, -1):
        first[l], c = c, (c + count[l]) >> 1
    code = {}
    for s in sorted(length):
        code[s] = first[length[s]]
        first[length[s]] += 1
    # the packed table
    table = BitWriter()
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            run = 1
            while s + run <= iM and run < LONGEST_LONG_RUN and length.get(s + run, 0) == 0:
                run += 1
            if run >= 2:
                if run >= SHORTEST_LONG_RUN:
                    table.put(6, LONG_ZEROCODE_RUN)
                    table.put(8, run - SHORTEST_LONG_RUN)
                else:
                    table.put(6, SHORT_ZEROCODE_RUN + run - 2)
                s += run
                continue
        table.put(6, l)
        s += 1
    table_bytes = table.flush()
    # the data: a symbol, or symbol + run-length symbol + 8-bit count where that is shorter
    data = BitWriter()

    def send(symbol, repeats):
        ls, lr = length[symbol], length[iM]
        if ls + lr + 8 < ls * repeats:
            data.put(ls, code[symbol])
            data.put(lr, code[iM])
            data.put(8, repeats)
        else:
            for _ in range(repeats + 1):
                data.put(ls, code[symbol])

    prev, repeats = words[0], 0
    for s in words[1:]:
        if s == prev and repeats < 255:
            repeats += 1
        else:
            send(prev, repeats)
            repeats = 0
        prev = s
    send(prev, repeats)
    n_bits = data.bits
    data_bytes = data.flush()
    return struct.pack("<5I", im, iM, len(table_bytes), n_bits, 0) + table_bytes + data_bytes


def piz_chunk(channel_planes):
    """channel_planes: list of uint16 arrays [rows, width * words_per_pixel], one per channel -> the PIZ chunk bytes"""
    flat = np.concatenate([p.reshape(-1) for p in channel_planes]).astype(np.uint16)
    present = np.zeros(1 << 16, bool)
    present[flat] = True
    present[0] = False  # zero is implicit: never in the bitmap
    nz = np.flatnonzero(present)
    bitmap = np.zeros(8192, np.uint8)
    for v in nz:
        bitmap[v >> 3] |= 1 << (v & 7)
    used = np.flatnonzero(bitmap)
    if len(used):
        min_nz, max_nz = int(used[0]), int(used[-1])
        head = struct.pack("<HH", min_nz, max_nz) + bitmap[min_nz:max_nz + 1].tobytes()
    else:
        head = struct.pack("<HH", 8191, 0)  # min > max: no bitmap bytes
    present[0] = True
    forward = np.cumsum(present) - 1
    max_value = int(forward[-1])
    words = []
    for plane in channel_planes:
        rows, n = plane.shape
        per = n // plane_width(plane)
        q = forward[plane.astype(np.int64)].astype(np.int64)
        for j in range(per):  # a 32-bit channel is two interleaved 16-bit planes
            sub = q[:, j::per].copy()
            wavelet_encode(sub, max_value < (1 << 14))
            q[:, j::per] = sub
        words.extend(int(v) for v in q.reshape(-1))
    huf = huffman_compress(words)
    return head + struct.pack("<i", len(huf)) + huf


_WIDTH = {}


def plane_width(plane):
    return _WIDTH[id(plane)]


def write_exr_piz(path, image, half=True, channels="RGB", store_raw_when_larger=False):
    """image: float array [h, w, len(channels)] -> single-part scanline EXR, PIZ, HALF or FLOAT channels.
    Returns the number of chunks that were PIZ coded (a writer stores a chunk raw when coding does not shrink it; off by default
    here so that noise exercises the coder too -- a reader tells the two apart by the chunk size alone)."""
    image = np.asarray(image, np.float32)
    h, w, n = image.shape
    assert n == len(channels)
    order = sorted(range(n), key=lambda i: channels[i])  # channels are stored in alphabetical order
    header = bytearray(struct.pack("<II", 20000630, 2))

    def attr(name, kind, payload):
        header.extend(name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<I", len(payload)) + payload)

    chlist = b"".join(channels[i].encode() + b"\0" + struct.pack("<IB3xII", 1 if half else 2, 0, 1, 1) for i in order) + b"\0"
    attr("channels", "chlist", chlist)
    attr("compression", "compression", bytes([4]))
    attr("dataWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, w - 1, h - 1))
    attr("lineOrder", "lineOrder", bytes([0]))
    attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    header.append(0)
    chunks = []
    coded = (h + 31) // 32
    for y0 in range(0, h, 32):
        rows = min(32, h - y0)
        planes = []
        for i in order:
            block = image[y0:y0 + rows, :, i]
            if half:
                plane = block.astype(np.float16).view(np.uint16).reshape(rows, w)
            else:
                plane = np.ascontiguousarray(block).view(np.uint16).reshape(rows, w * 2)
            plane = np.ascontiguousarray(plane)
            _WIDTH[id(plane)] = w
            planes.append(plane)
        body = piz_chunk(planes)
        raw_size = sum(p.size for p in planes) * 2
        if len(body) == raw_size or (store_raw_when_larger and len(body) > raw_size):  # stored raw, in scanline order
            coded -= 1
            body = b"".join(planes[c][r].tobytes() for r in range(rows) for c in range(len(planes)))
        chunks.append(struct.pack("<iI", y0, len(body)) + body)
    offset = len(header) + 8 * len(chunks)
    table = bytearray()
    for c in chunks:
        table.extend(struct.pack("<Q", offset))
        offset += len(c)
    with open(path, "wb") as f:
        f.write(bytes(header) + bytes(table) + b"".join(chunks))
    return coded
