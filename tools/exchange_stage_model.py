"""Cheetah and Lion encoding as PASSES OF BLIND EXCHANGES, restated in Python — the formulation behind
density_amd/csrc/exchange_stages.hip, checked against the oracle by tests/test_exchange_stage_model.py.

The reference walks a stream quad by quad through three tables (cheetah.rs:123-149) or seven (lion.rs:211-270).  Every one of those
table updates is an unconditional EXCHANGE once the quads that take part are known:

  Cheetah   P: old = xchg(pred[h(q[i-1])], q[i])     predicted  <=> old == q[i]        (a predicted quad rewrites what is there)
            A: old = xchg(a[h(q[i])], q[i])          for the others;   MAP_A <=> old == q[i]
            B: old = xchg(b[h(q[i])], old_A)         for the rest;     MAP_B <=> old == q[i]      (cheetah.rs:140-141: b = a, a = quad)
  Lion      P0..P4: the five-deep move-to-front list per slot (lion.rs:50-57, 240-262) is a chain of five such exchanges, each
            handing the value it displaced to the next level and stopping where it finds the quad; then A and B as above.

so a whole stage can run over the whole stream before the next one starts ("pass-major"), each stage being "one table, every
taking-part quad in stream order" — on the GPU an ordered LDS exchange per 64 quads, the table cut by key halves to fit the LDS.
No raw-copy blocks are assumed (codec.rs:35-37 would take their quads out of every table): the caller checks the record sizes
afterwards and falls back where two incompressible records meet.
"""
import numpy as np

M = 0x9D6EF916


def hashes(q):
    return ((q.astype(np.uint64) * M) & 0xFFFFFFFF).astype(np.uint32) >> 16


def stage(keys, values, taking_part, probe, table=None):
    """One table (zero-initialised unless given); in stream order every taking-part quad exchanges `values[i]` into slot `keys[i]`.
    Returns (old values, hit = old == probe[i])."""
    table = np.zeros(65536, dtype=np.uint32) if table is None else table.copy()
    old = np.zeros(len(keys), dtype=np.uint32)
    for i in np.nonzero(taking_part)[0]:
        k = keys[i]
        old[i] = table[k]
        table[k] = values[i]
    return old, taking_part & (old == probe)


def cheetah_flags(q, last_hash=0, tables=(None, None, None)):
    h = hashes(q)
    prev = np.concatenate(([last_hash], h[:-1])).astype(np.uint32)    # last_hash starts at 0 (cheetah.rs:52) and runs through the blocks
    everyone = np.ones(len(q), dtype=bool)
    _, predicted = stage(prev, q, everyone, q, tables[0])
    old_a, map_a = stage(h, q, ~predicted, q, tables[1])
    _, map_b = stage(h, old_a, ~predicted & ~map_a, q, tables[2])
    flags = np.zeros(len(q), dtype=np.uint8)                           # 0 plain
    flags[map_a] = 1
    flags[map_b] = 2
    flags[predicted] = 3
    return flags, h


def lion_flags(q, last_hash=0, tables=(None,) * 7):
    h = hashes(q)
    prev = np.concatenate(([last_hash], h[:-1])).astype(np.uint32)
    left = np.ones(len(q), dtype=bool)
    flags = np.zeros(len(q), dtype=np.uint8)
    carry = q
    for level in range(5):                                             # the list: level k takes what level k-1 displaced
        carry_old, hit = stage(prev, carry, left, q, tables[level])
        flags[hit] = level + 1
        left = left & ~hit
        carry = carry_old
    old_a, map_a = stage(h, q, left, q, tables[5])
    _, map_b = stage(h, old_a, left & ~map_a, q, tables[6])
    flags[map_a] = 6
    flags[map_b] = 7
    return flags, h


def assemble(algo, data, flags, h):
    """The stream of whole records those flags make (no raw copies, no ragged end): signature, then the items."""
    bits, per = (2, 32) if algo == "cheetah" else (3, 16)
    sig_bytes = per * bits // 8
    plain = 0
    two_bytes = (1, 2) if algo == "cheetah" else (6, 7)
    q = np.frombuffer(data, dtype="<u4")
    out = bytearray()
    sizes = []
    for r in range(len(q) // per):
        sig = 0
        items = bytearray()
        for k in range(per):
            i = r * per + k
            f = int(flags[i])
            sig |= f << (bits * k)
            if f == plain:
                items += int(q[i]).to_bytes(4, "little")
            elif f in two_bytes:
                items += int(h[i]).to_bytes(2, "little")
        out += sig.to_bytes(sig_bytes, "little") + items
        sizes.append(sig_bytes + len(items))
    return bytes(out), sizes


def encode(algo, data):
    """-> (stream, record sizes) for `data` of whole blocks, assuming no raw-copy block is ever taken."""
    q = np.frombuffer(data, dtype="<u4")
    flags, h = cheetah_flags(q) if algo == "cheetah" else lion_flags(q)
    return assemble(algo, data, flags, h)


def cheetah_head_in_order(data, head_bytes):
    """The first head_bytes of a chunk the reference's way — tables, blow-up protection (protection_state.rs:19-47) and all — in plain
    Python: -> (stream, tables P/A/B, last_hash, penalty running, last record incompressible).  What the one-wave kernel does for the
    cold-dictionary start of every chunk before the passes take over."""
    q = np.frombuffer(data[:head_bytes], dtype="<u4")
    P = np.zeros(65536, dtype=np.uint32); A = P.copy(); B = P.copy()
    out = bytearray()
    penalty, start, prev_inc, counter, last_hash = 0, 1, False, 0, 0
    for r in range(len(q) // 32):
        if (counter & 0xF) == 0 and start > 1:
            start >>= 1
        counter += 1
        block = q[32 * r: 32 * r + 32]
        if penalty > 0:
            out += block.tobytes()
            penalty -= 1
            if penalty == 0:
                start += 1
            continue
        sig, items = 0, bytearray()
        for k, x in enumerate(block):
            x = int(x)
            hx = (x * M & 0xFFFFFFFF) >> 16
            if P[last_hash] == x:
                f = 3
            else:
                if A[hx] == x:
                    f = 1
                    items += hx.to_bytes(2, "little")
                else:
                    if B[hx] == x:
                        f = 2
                        items += hx.to_bytes(2, "little")
                    else:
                        f = 0
                        items += x.to_bytes(4, "little")
                    B[hx] = A[hx]
                    A[hx] = x
                P[last_hash] = x
            last_hash = hx
            sig |= f << (2 * k)
        out += sig.to_bytes(8, "little") + items
        inc = 8 + len(items) >= 128
        if inc and prev_inc:
            penalty = start
        prev_inc = inc
    return bytes(out), (P, A, B), last_hash, penalty > 0, prev_inc


def cheetah_encode_head_then_passes(data, head_bytes):
    """-> the chunk's stream, or None where the passes must hand the chunk back (the head ends inside a penalty, or two incompressible
    records meet behind it)."""
    head, tables, last_hash, in_penalty, prev_inc = cheetah_head_in_order(data, head_bytes)
    if in_penalty:
        return None
    rest = data[head_bytes:]
    q = np.frombuffer(rest, dtype="<u4")
    flags, h = cheetah_flags(q, last_hash, tables)
    body, sizes = assemble("cheetah", rest, flags, h)
    inc = [prev_inc] + [s >= 128 for s in sizes]
    if any(a and b for a, b in zip(inc, inc[1:])):
        return None
    return head + body


def lion_head_in_order(data, head_bytes):
    """As cheetah_head_in_order for Lion (lion.rs:211-270): -> (stream, tables N0..N4/A/B, last_hash, penalty running, last record incompressible)."""
    q = np.frombuffer(data[:head_bytes], dtype="<u4")
    N = [np.zeros(65536, dtype=np.uint32) for _ in range(5)]
    A = np.zeros(65536, dtype=np.uint32); B = A.copy()
    out = bytearray()
    penalty, start, prev_inc, counter, last_hash = 0, 1, False, 0, 0
    for r in range(len(q) // 16):
        if (counter & 0xF) == 0 and start > 1:
            start >>= 1
        counter += 1
        block = q[16 * r: 16 * r + 16]
        if penalty > 0:
            out += block.tobytes()
            penalty -= 1
            if penalty == 0:
                start += 1
            continue
        sig, items = 0, bytearray()
        for k, x in enumerate(block):
            x = int(x)
            hx = (x * M & 0xFFFFFFFF) >> 16
            row = [int(n[last_hash]) for n in N]
            if x in row:
                depth = row.index(x)
                f = depth + 1
            else:
                depth = 4                                               # shift_predictions: everything moves down, the last entry falls out
                if A[hx] == x:
                    f = 6
                    items += hx.to_bytes(2, "little")
                else:
                    if B[hx] == x:
                        f = 7
                        items += hx.to_bytes(2, "little")
                    else:
                        f = 0
                        items += x.to_bytes(4, "little")
                    B[hx] = A[hx]
                    A[hx] = x
            for i in range(depth, 0, -1):                               # move to front (lion.rs:50-57, 240-262)
                N[i][last_hash] = row[i - 1]
            N[0][last_hash] = x
            last_hash = hx
            sig |= f << (3 * k)
        out += sig.to_bytes(6, "little") + items
        inc = 6 + len(items) >= 64
        if inc and prev_inc:
            penalty = start
        prev_inc = inc
    return bytes(out), tuple(N) + (A, B), last_hash, penalty > 0, prev_inc


def encode_head_then_passes(algo, data, head_bytes):
    """-> the chunk's stream, or None where the passes must hand the chunk back (the head ends inside a penalty, or two incompressible
    records meet behind it)."""
    head_fn, flags_fn, rec = (cheetah_head_in_order, cheetah_flags, 128) if algo == "cheetah" else (lion_head_in_order, lion_flags, 64)
    head, tables, last_hash, in_penalty, prev_inc = head_fn(data, head_bytes)
    if in_penalty:
        return None
    rest = data[head_bytes:]
    q = np.frombuffer(rest, dtype="<u4")
    flags, h = flags_fn(q, last_hash, tables)
    body, sizes = assemble(algo, rest, flags, h)
    inc = [prev_inc] + [s >= rec for s in sizes]
    if any(a and b for a, b in zip(inc, inc[1:])):
        return None
    return head + body
