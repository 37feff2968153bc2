"""Second, independently structured CPU model of the density stream format — TEST INFRASTRUCTURE ONLY.

Pure Python, written from the format description rather than as a line-by-line
restatement: Chameleon uses the *position* formulation (dict[h] == the most recent quad with hash h inside a
coded block), Cheetah/Lion use small list-based LRU models.  It exists to be diffed against
oracle/density_oracle.c on random and adversarial inputs (tests/test_oracle_differential.py), because the
reference's own tests pin only one 125-byte input (src/lib.rs:19-72).  Small inputs only (it is slow).
"""
import struct

MUL = 0x9D6EF916
GEOM = {  # algo: (flag_bits, sig_bytes, block_bytes)   chameleon.rs:138-146, cheetah.rs:188-196, lion.rs:317-325
    "chameleon": (1, 8, 256),
    "cheetah": (2, 8, 128),
    "lion": (3, 6, 64),
}


def h16(q):
    return ((q * MUL) & 0xFFFFFFFF) >> 16


class Guard:
    """codec/protection_state.rs:1-47"""

    def __init__(self):
        self.penalty, self.start, self.prev, self.count = 0, 1, False, 0

    def next_is_copy(self):
        if self.count % 16 == 0 and self.start > 1:
            self.start //= 2
        self.count += 1
        return self.penalty > 0

    def decay(self):
        self.penalty -= 1
        if self.penalty == 0:
            self.start += 1

    def update(self, incompressible):
        if incompressible and self.prev:
            self.penalty = self.start
        self.prev = incompressible


class _Cheetah:
    def __init__(self):
        self.pairs, self.follow, self.last = {}, {}, 0

    def step(self, q):
        """-> (flag, item bytes)"""
        h = h16(q)
        if self.follow.get(self.last, 0) == q:
            self.last = h
            return 3, b""
        lru = self.pairs.setdefault(h, [0, 0])
        if lru[0] == q:
            flag, item = 1, struct.pack("<H", h)
        else:
            flag, item = (2, struct.pack("<H", h)) if lru[1] == q else (0, struct.pack("<I", q))
            lru[1], lru[0] = lru[0], q
        self.follow[self.last] = q
        self.last = h
        return flag, item

    def unstep(self, flag, rd):
        if flag == 3:
            q = self.follow.get(self.last, 0)
            h = h16(q)
        else:
            if flag == 0:
                q = rd.u32(); h = h16(q)
                lru = self.pairs.setdefault(h, [0, 0]); lru[1], lru[0] = lru[0], q
            else:
                h = rd.u16(); lru = self.pairs.setdefault(h, [0, 0])
                if flag == 1:
                    q = lru[0]
                else:
                    q = lru[1]; lru[1], lru[0] = lru[0], q
            self.follow[self.last] = q
        self.last = h
        return q


class _Lion:
    def __init__(self):
        self.pairs, self.follow, self.last = {}, {}, 0

    def _f(self):
        return self.follow.setdefault(self.last, [0, 0, 0, 0, 0])

    def step(self, q):
        h = h16(q)
        f = self._f()
        if q in f:
            k = f.index(q)               # first match wins: A before B ... before E
            if k:
                del f[k]; f.insert(0, q)  # move to front (lion.rs:240-262)
            self.last = h
            return k + 1, b""
        lru = self.pairs.setdefault(h, [0, 0])
        if lru[0] == q:
            flag, item = 6, struct.pack("<H", h)
        else:
            flag, item = (7, struct.pack("<H", h)) if lru[1] == q else (0, struct.pack("<I", q))
            lru[1], lru[0] = lru[0], q
        f.insert(0, q); f.pop()
        self.last = h
        return flag, item

    def unstep(self, flag, rd):
        f = self._f()
        if 1 <= flag <= 5:
            q = f[flag - 1]
            if flag > 1:
                del f[flag - 1]; f.insert(0, q)
            h = h16(q)
        else:
            if flag == 0:
                q = rd.u32(); h = h16(q)
                lru = self.pairs.setdefault(h, [0, 0]); lru[1], lru[0] = lru[0], q
            else:
                h = rd.u16(); lru = self.pairs.setdefault(h, [0, 0])
                if flag == 6:
                    q = lru[0]
                else:
                    q = lru[1]; lru[1], lru[0] = lru[0], q
            f.insert(0, q); f.pop()
        self.last = h
        return q


class _ChameleonPos:
    """Position formulation: remember WHERE the last quad with each hash sat, compare against the input there."""

    def __init__(self, data):
        self.data, self.where = data, {}

    def step_at(self, pos, q):
        h = h16(q)
        prev = self.where.get(h)
        seen = struct.unpack_from("<I", self.data, prev)[0] if prev is not None else 0
        self.where[h] = pos
        return (1, struct.pack("<H", h)) if seen == q else (0, struct.pack("<I", q))


class _ChameleonDec:
    def __init__(self):
        self.d = {}

    def unstep(self, flag, rd):
        if flag == 0:
            q = rd.u32(); self.d[h16(q)] = q
            return q
        return self.d.get(rd.u16(), 0)


class _Rd:
    def __init__(self, b):
        self.b, self.i = b, 0

    def left(self):
        return len(self.b) - self.i

    def take(self, n):
        if n > self.left():
            raise ValueError("truncated stream")
        v = self.b[self.i:self.i + n]; self.i += n
        return v

    def u16(self):
        return struct.unpack("<H", self.take(2))[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]


def encode(algo, data):
    data = bytes(data)
    fbits, S, B = GEOM[algo]
    g = Guard()
    out = bytearray()
    coder = _ChameleonPos(data) if algo == "chameleon" else _Cheetah() if algo == "cheetah" else _Lion()
    copied = 0
    for pos in range(0, len(data), B):
        blk = data[pos:pos + B]
        if g.next_is_copy():
            out += blk; g.decay(); copied += 1
            continue
        sig, shift, items = 0, 0, bytearray()
        nq = len(blk) // 4
        for k in range(nq):
            q = struct.unpack_from("<I", blk, 4 * k)[0]
            flag, item = coder.step_at(pos + 4 * k, q) if algo == "chameleon" else coder.step(q)
            sig |= flag << shift; shift += fbits
            items += item
        items += blk[4 * nq:]
        out += sig.to_bytes(8, "little")[:S] + items
        g.update(S + len(items) >= B)
    return bytes(out), copied


def decode(algo, enc):
    fbits, S, B = GEOM[algo]
    g = Guard()
    rd = _Rd(bytes(enc))
    out = bytearray()
    dec = _ChameleonDec() if algo == "chameleon" else _Cheetah() if algo == "cheetah" else _Lion()
    mask = (1 << fbits) - 1
    while rd.left() > 0:
        if g.next_is_copy():
            n = min(B, rd.left())
            out += rd.take(n)
            if rd.left() == 0:
                break
            g.decay()
            continue
        mark = rd.i
        sig = int.from_bytes(rd.take(S), "little")
        done = False
        for _ in range(B // 4):
            flag = sig & mask; sig >>= fbits
            if flag == 0 and rd.left() < 4:
                out += rd.take(rd.left()); done = True
                break
            out += struct.pack("<I", dec.unstep(flag, rd))
        if done:
            break
        g.update(rd.i - mark >= B)
    return bytes(out)
