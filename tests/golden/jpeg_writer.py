"""A small JPEG encoder (test tooling): baseline and progressive DCT frames with chosen
sampling factors, restart intervals, interleaved or one-component scans, optional JFIF /
Adobe segments.  It exists to make input files for the texture reader's tests; what the
reference's loader (stb_image) makes of each file is recorded beside it by make_fixtures.py.
Component samples are taken as given (no colour conversion): the decoder under test does that.
"""
import heapq
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


def _dct_matrix():
    k = np.arange(8)
    m = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    m[0] /= np.sqrt(2)
    return m


DCT = _dct_matrix()


def _code_lengths(freq):
    """Huffman code lengths (at most 16, no all-ones code) for the symbols of freq, as T.81 K.2 does it."""
    symbols = sorted(freq)
    heap = [(freq[s], i, [s]) for i, s in enumerate(symbols)] + [(0, len(symbols), [256])]  # 256 reserves the all-ones code
    heapq.heapify(heap)
    depth = {s: 0 for s in symbols + [256]}
    tie = len(heap)
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        for s in a[2] + b[2]:
            depth[s] += 1
        heapq.heappush(heap, (a[0] + b[0], tie, a[2] + b[2]))
        tie += 1
    bits = [0] * 40
    for s in depth:
        bits[max(depth[s], 1)] += 1
    i = 39
    while i > 16:
        while bits[i] > 0:
            j = i - 2
            while bits[j] == 0:
                j -= 1
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
        i -= 1
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1  # the reserved symbol
    order = sorted(symbols, key=lambda s: (depth[s], -freq[s], s))
    return bits[1:17], order


class _Table:
    def __init__(self, freq):
        self.counts, self.order = _code_lengths(freq)
        self.code = {}
        code, k = 0, 0
        for length in range(1, 17):
            for _ in range(self.counts[length - 1]):
                self.code[self.order[k]] = (code, length)
                code += 1
                k += 1
            code <<= 1

    def segment(self, cls, ident):
        body = bytes([cls << 4 | ident]) + bytes(self.counts) + bytes(self.order)
        return b"\xff\xc4" + struct.pack(">H", len(body) + 2) + body


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, n):
        self.acc = (self.acc << n) | (value & ((1 << n) - 1))
        self.n += n
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 255:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _magnitude(v):
    """(size, bits) of a coefficient difference / value."""
    a = abs(v)
    size = a.bit_length()
    return size, (v if v >= 0 else v - 1) & ((1 << size) - 1)


class _Scan:
    """Tokens of one scan: ('sym', (class, table), symbol), ('bits', value, n), ('rst', k)."""

    def __init__(self):
        self.tokens = []

    def sym(self, kind, s):
        self.tokens.append(("sym", kind, s))

    def bits(self, v, n):
        if n:
            self.tokens.append(("bits", v, n))


def write_jpeg(path, samples, sampling=((1, 1),), quality_tables=None, progressive=False, restart=0,
               interleaved=True, jfif=True, adobe=None, ids=None, scans=None, seed=0, filler=False):
    """samples: (h, w, ncomp) uint8 component samples at full resolution (subsampled here by
    averaging).  sampling: (h, v) per component.  scans (progressive): list of
    (components, ss, se, ah, al); a default script is used when None."""
    samples = np.asarray(samples)
    H, W, n = samples.shape
    sampling = list(sampling) * (n if len(sampling) == 1 else 1)
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    rng = np.random.default_rng(seed)
    if quality_tables is None:
        quality_tables = [np.clip(rng.integers(1, 24, 64) + np.arange(64) // 4, 1, 255) for _ in range(2)]
    ids = ids or list(range(1, n + 1))
    # coefficient blocks per component: (blocksY, blocksX, 64) in zigzag order
    comps = []
    for c in range(n):
        h, v = sampling[c]
        fx, fy = hmax // h, vmax // v
        cw, ch = -(-W * h // hmax), -(-H * v // vmax)
        plane = samples[:, :, c].astype(float)
        plane = np.pad(plane, ((0, ch * fy - H), (0, cw * fx - W)), mode="edge")
        plane = plane.reshape(ch, fy, cw, fx).mean(axis=(1, 3))
        sw, sh = mx * h * 8, my * v * 8
        plane = np.pad(plane, ((0, sh - ch), (0, sw - cw)), mode="edge") - 128.0
        q = quality_tables[0 if c == 0 else 1].astype(float)
        blocks = plane.reshape(sh // 8, 8, sw // 8, 8).transpose(0, 2, 1, 3)
        coef = np.einsum("ij,abjk,lk->abil", DCT, blocks, DCT).reshape(sh // 8, sw // 8, 64)[:, :, ZIGZAG]
        coef = np.rint(coef / q[None, None, :]).astype(int)
        comps.append(dict(h=h, v=v, tq=0 if c == 0 else 1, coef=coef, bw=-(-cw // 8), bh=-(-ch // 8)))

    def blocks_of(scan_comps):
        """Block coordinates in coding order, with MCU boundaries: yields lists of (comp, by, bx)."""
        if len(scan_comps) == 1:
            c = scan_comps[0]
            for by in range(comps[c]["bh"]):
                for bx in range(comps[c]["bw"]):
                    yield [(c, by, bx)]
        else:
            for j in range(my):
                for i in range(mx):
                    yield [(c, j * comps[c]["v"] + y, i * comps[c]["h"] + x) for c in scan_comps
                           for y in range(comps[c]["v"]) for x in range(comps[c]["h"])]

    def encode_scan(scan_comps, ss, se, ah, al):
        sc = _Scan()
        pred = {c: 0 for c in scan_comps}
        state = dict(eobrun=0, be=[])
        kind_dc = {c: ("dc", 0 if c == 0 else 1) for c in scan_comps}
        kind_ac = {c: ("ac", 0 if c == 0 else 1) for c in scan_comps}

        def emit_eobrun(c):
            if state["eobrun"]:
                nb = state["eobrun"].bit_length() - 1
                sc.sym(kind_ac[c], nb << 4)
                sc.bits(state["eobrun"] & ((1 << nb) - 1), nb)
                state["eobrun"] = 0
            for b in state["be"]:
                sc.bits(b, 1)
            state["be"] = []

        todo = restart
        rst = 0
        for mcu in blocks_of(scan_comps):
            for c, by, bx in mcu:
                z = comps[c]["coef"][by, bx]
                if not progressive:
                    d = int(z[0]) - pred[c]
                    pred[c] = int(z[0])
                    size, bits = _magnitude(d)
                    sc.sym(kind_dc[c], size)
                    sc.bits(bits, size)
                    run = 0
                    last = max([k for k in range(1, 64) if z[k]], default=0)
                    for k in range(1, last + 1):
                        if z[k] == 0:
                            run += 1
                            continue
                        while run > 15:
                            sc.sym(kind_ac[c], 0xF0)
                            run -= 16
                        size, bits = _magnitude(int(z[k]))
                        sc.sym(kind_ac[c], run << 4 | size)
                        sc.bits(bits, size)
                        run = 0
                    if last < 63:
                        sc.sym(kind_ac[c], 0)
                elif ss == 0:
                    v = int(z[0]) >> al
                    if ah == 0:
                        size, bits = _magnitude(v - pred[c])
                        pred[c] = v
                        sc.sym(kind_dc[c], size)
                        sc.bits(bits, size)
                    else:
                        sc.bits(v & 1, 1)
                elif ah == 0:
                    run = 0
                    for k in range(ss, se + 1):
                        a = abs(int(z[k])) >> al
                        if a == 0:
                            run += 1
                            continue
                        emit_eobrun(c)
                        while run > 15:
                            sc.sym(kind_ac[c], 0xF0)
                            run -= 16
                        size = a.bit_length()
                        sc.sym(kind_ac[c], run << 4 | size)
                        sc.bits(a if z[k] > 0 else (~a) & ((1 << size) - 1), size)
                        run = 0
                    if run:
                        state["eobrun"] += 1
                        if state["eobrun"] == 0x7FFF:
                            emit_eobrun(c)
                else:
                    absv = [abs(int(z[k])) >> al for k in range(64)]
                    eob = max([k for k in range(ss, se + 1) if absv[k] == 1], default=-1)
                    run, br = 0, []
                    for k in range(ss, se + 1):
                        a = absv[k]
                        if a == 0:
                            run += 1
                            continue
                        while run > 15 and k <= eob:
                            emit_eobrun(c)
                            sc.sym(kind_ac[c], 0xF0)
                            run -= 16
                            for b in br:
                                sc.bits(b, 1)
                            br = []
                        if a > 1:
                            br.append(a & 1)
                            continue
                        emit_eobrun(c)
                        sc.sym(kind_ac[c], run << 4 | 1)
                        sc.bits(0 if z[k] < 0 else 1, 1)
                        for b in br:
                            sc.bits(b, 1)
                        br = []
                        run = 0
                    if run or br:
                        state["eobrun"] += 1
                        state["be"] += br
                        if state["eobrun"] == 0x7FFF or len(state["be"]) > 900:
                            emit_eobrun(c)
            if restart:
                todo -= 1
                if todo == 0:
                    emit_eobrun(scan_comps[0])
                    sc.tokens.append(("rst", rst))
                    rst = (rst + 1) & 7
                    todo = restart
                    pred = {c: 0 for c in scan_comps}
        emit_eobrun(scan_comps[0])
        if sc.tokens and sc.tokens[-1][0] == "rst":
            sc.tokens.pop()  # no restart marker after the last MCU
        return sc

    if not progressive:
        script = [(list(range(n)), 0, 63, 0, 0)] if interleaved and n > 1 else [([c], 0, 63, 0, 0) for c in range(n)]
    elif scans is not None:
        script = scans
    else:
        script = [(list(range(n)) if interleaved else [0], 0, 0, 0, 1)]
        if not interleaved:
            script += [([c], 0, 0, 0, 1) for c in range(1, n)]
        script += [([0], 1, 5, 0, 2)]
        script += [([c], 1, 63, 0, 1) for c in range(1, n)]
        script += [([0], 6, 63, 0, 2), ([0], 1, 63, 2, 1), (list(range(n)) if interleaved else [0], 0, 0, 1, 0)]
        if not interleaved:
            script += [([c], 0, 0, 1, 0) for c in range(1, n)]
        script += [([c], 1, 63, 1, 0) for c in range(n - 1, -1, -1)]

    out = bytearray(b"\xff\xd8")
    if jfif:
        out += b"\xff\xe0" + struct.pack(">H", 16) + b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0"
    if filler:
        out += b"\xff\xfe" + struct.pack(">H", 2 + 11) + b"hello world" + b"\xff\xff"   # a comment, then fill bytes before the next marker
    if adobe is not None:
        out += b"\xff\xee" + struct.pack(">H", 14) + b"Adobe\0" + bytes([100, 0, 0, 0, 0, adobe])
    for t, q in enumerate(quality_tables):
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([t]) + bytes(int(v) for v in q)
    out += (b"\xff\xc2" if progressive else b"\xff\xc0") + struct.pack(">HBHHB", 8 + 3 * n, 8, H, W, n)
    for c in range(n):
        out += bytes([ids[c], comps[c]["h"] << 4 | comps[c]["v"], comps[c]["tq"]])
    if restart:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart)
    for scan_comps, ss, se, ah, al in script:
        sc = encode_scan(scan_comps, ss, se, ah, al)
        per = {}
        for t in sc.tokens:
            if t[0] == "sym":
                per.setdefault(t[1], {}).setdefault(t[2], 0)
                per[t[1]][t[2]] += 1
        tables = {}
        for kind, fr in sorted(per.items()):
            tables[kind] = _Table(fr)
            out += tables[kind].segment(1 if kind[0] == "ac" else 0, kind[1])
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(scan_comps), len(scan_comps))
        for c in scan_comps:
            t = 0 if c == 0 else 1
            out += bytes([ids[c], t << 4 | t])
        out += bytes([ss, se, ah << 4 | al])
        bw = _Bits()
        for t in sc.tokens:
            if t[0] == "bits":
                bw.put(t[1], t[2])
            elif t[0] == "rst":
                bw.flush()
                bw.out += bytes([0xFF, 0xD0 + t[1]])
            else:
                code, length = tables[t[1]].code[t[2]]
                bw.put(code, length)
        bw.flush()
        out += bw.out
    out += b"\xff\xd9"
    with open(path, "wb") as fh:
        fh.write(bytes(out))
