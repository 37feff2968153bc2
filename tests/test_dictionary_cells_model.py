"""The decode passes' dictionary (density_amd/csrc/decode_passes.hip, `cheetah_pass<1>`): chunk_map[h] = {a, b} kept as two cells X, Y and an order bit o
(a = o ? Y : X).  PLAIN writes its quad to the cell b sits in and toggles o; MAP_B reads the cell b sits in and toggles o; MAP_A reads the cell a sits in —
which makes o a function of the FLAGS alone (an ordered XOR per slot) and the cells plain last-writer cells (an ordered exchange per cell).  Held here against
cheetah.rs:69-93 (decode_plain / decode_map_a / decode_map_b; Lion's lion.rs:85-124 are the same three) on random operation streams, including MAP reads of
slots nothing has written."""
import random

import pytest


def reference(ops):
    """ops: (flag, slot, quad-for-PLAIN) -> the quads MAP operations return; flags 0 PLAIN, 1 MAP_A, 2 MAP_B"""
    table, out = {}, []
    for f, s, q in ops:
        a, b = table.get(s, (0, 0))
        if f == 0:
            table[s] = (q, a); out.append(q)                  # cheetah.rs:69-75
        elif f == 1:
            out.append(a)                                     # :78-83
        else:
            table[s] = (b, a); out.append(b)                  # :85-93: swap
    return out


def cells(ops, dense_cap=None):
    """the passes: group A gives every operation the order bit it meets (XOR), group B touches one cell; `dense_cap`: the taking-part operations of a trip of
    1024 packed first (the order is kept: what the dense form of the kernel does), everything else left out"""
    o_bits, cell, out = {}, {}, []
    met = []
    for f, s, q in ops:                                       # group A, in order
        o = o_bits.get(s, 0)
        met.append(o)
        if f != 1:
            o_bits[s] = o ^ 1
    for (f, s, q), o in zip(ops, met):                        # group B, in order
        mine = o if f == 1 else 1 - o                         # a sits in cell o, b in cell 1 - o
        if f == 0:
            cell[(s, mine)] = q; out.append(q)
        else:
            out.append(cell.get((s, mine), 0))
    return out


@pytest.mark.parametrize("n_slots,p_plain", [(1, 0.3), (3, 0.5), (40, 0.4), (5000, 0.3)])
def test_cells_and_order_bit_are_the_pair(n_slots, p_plain):
    for seed in range(20):
        rnd = random.Random(seed)
        ops = []
        for _ in range(3000):
            r = rnd.random()
            f = 0 if r < p_plain else (1 if r < p_plain + (1 - p_plain) * 0.6 else 2)
            ops.append((f, rnd.randrange(n_slots), rnd.randrange(1, 1 << 32)))
        assert cells(ops) == reference(ops), (seed, n_slots)


def test_reads_of_slots_nothing_has_written_return_zero():
    assert cells([(1, 5, 0), (2, 5, 0), (1, 5, 0), (0, 5, 77), (2, 5, 0), (1, 5, 0)]) == reference([(1, 5, 0), (2, 5, 0), (1, 5, 0), (0, 5, 77), (2, 5, 0), (1, 5, 0)]) == [0, 0, 0, 77, 0, 0]
