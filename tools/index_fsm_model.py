"""The rotation decoder's check of a block index's raw-copy bits (rotor.hip::index_fsm_consistent), restated in Python, next to the
plain walk of the blow-up protection FSM it must be equivalent to (protection_state.rs:19-47, codec.rs:89-91).

An index entry: bit 7 = raw-copy block, bits 0..6 = MAP count of a coded block (0x7f: the ragged last block).  A coded block is
incompressible iff its record is 256 bytes or more: at most 4 MAP flags."""


def fsm_walk_ok(ix):
    """Ground truth: run the FSM over the index; every raw bit must be what the FSM decides."""
    penalty, start, prev, counter = 0, 1, False, 0
    for e in ix:
        if (counter & 15) == 0 and start > 1:
            start >>= 1
        counter += 1
        is_copy = penalty > 0
        if bool(e & 0x80) != is_copy:
            return False
        if is_copy:
            penalty -= 1
            if penalty == 0:
                start = (start + 1) & 0xFF
        else:
            inc = e <= 4
            if inc and prev:
                penalty = start
            prev = inc
    return True


def _mult16(lo, hi):
    return hi // 16 + 1 - (lo + 15) // 16


def _halve(s, k):
    h = s >> k if k < 32 else 0
    return (h if h else 1) if s > 1 else s


def block_consistent(ix, i):
    """rotor.hip::index_fsm_consistent for block i (called where the block is raw or incompressible)."""
    n = len(ix)
    raw = lambda b: bool(ix[b] & 0x80)
    inc = lambda b: ix[b] <= 4
    if not raw(i):
        u = i
        while u > 0 and raw(u - 1):
            u -= 1
        trigger = u > 0 and inc(u - 1)
        return (not trigger) or i + 1 >= n or raw(i + 1)
    if i > 0 and raw(i - 1):
        return True
    if i == 0 or not inc(i - 1):
        return False
    t = i - 1
    u = t
    while u > 0 and raw(u - 1):
        u -= 1
    if u == 0 or not inc(u - 1):
        return False
    L = 1
    while i + L < n and raw(i + L):
        L += 1
    s = 1
    reach = t - 143 if t > 143 else 0
    e = t
    while e > reach and not raw(e - 1):
        e -= 1
    if e > reach:
        last = e - 1
        a = last
        while a > 0 and raw(a - 1) and last - a < 255:
            a -= 1
        s_end = (_halve(last - a + 1, _mult16(a, last)) + 1) & 0xFF
        s = _halve(s_end, _mult16(last + 1, t))
    return L == s or (L < s and i + L == n)


def index_consistent(ix):
    return all(block_consistent(ix, i) for i, e in enumerate(ix) if (e & 0x80) or e <= 4)


def make_index(incs):
    """The index the FSM produces for a sequence of 'would be incompressible if coded' flags (MAP counts 2 / 40)."""
    out = bytearray()
    penalty, start, prev, counter = 0, 1, False, 0
    for inc in incs:
        if (counter & 15) == 0 and start > 1:
            start >>= 1
        counter += 1
        if penalty > 0:
            out.append(0x80)
            penalty -= 1
            if penalty == 0:
                start = (start + 1) & 0xFF
        else:
            out.append(2 if inc else 40)
            if inc and prev:
                penalty = start
            prev = inc
    return out
