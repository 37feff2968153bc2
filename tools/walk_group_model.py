"""The grouped contexts walk of decode_passes.hip::cheetah_walk<NB> (speculate by reads, one ordered pass, verify, take back, go again) and the turn of its team
form cheetah_walk_team (speculative reads ahead of the turn against a stale table, lane 0 patched under the token, a wrong read costs its chain) restated in Python and held
against the sequential walk (cheetah.rs:72,81,90,97-102 on hashes) on adversarial descriptor streams.  tests/test_walk_group_model.py runs it."""
import random


def sequential(pred, h, hw, c0, H):
    """contexts per quad; H: dict context -> hash of the value last left there (default 0)"""
    ctx, c = [], c0
    for p, hi, wi in zip(pred, h, hw):
        ctx.append(c)
        if p:
            c = H.get(c, 0)
        else:
            H[c] = wi
            c = hi
    return ctx, c


def grouped(pred, h, hw, c0, H, G=2, L=64):
    """one group of G blocks of L lanes (len(pred) == G * L), exactly as the kernel does it; returns (contexts, next running context, passes)"""
    n = G * L
    assert len(pred) == n
    P = [[pred[b * L + i] for i in range(L)] for b in range(G)]
    cv = [[0] * L for _ in range(G)]
    known = [[False] * L for _ in range(G)]
    K0 = [[False] * L for _ in range(G)]
    for b in range(G):
        for i in range(L):
            if i == 0:
                K0[b][0] = True if b == 0 else (not P[b - 1][L - 1])
                cv[b][0] = c0 if b == 0 else h[(b - 1) * L + L - 1]
            else:
                K0[b][i] = not P[b][i - 1]
                cv[b][i] = h[b * L + i - 1]
            known[b][i] = K0[b][i]
    fin = [[False] * L for _ in range(G)]
    rs = [[0] * L for _ in range(G)]
    rf = [[0] * L for _ in range(G)]
    passes = 0
    while True:
        passes += 1
        rdone = [row[:] for row in fin]
        while True:
            R = [[P[b][i] and known[b][i] and not rdone[b][i] for i in range(L)] for b in range(G)]
            if not any(any(r) for r in R):
                break
            r = [[H.get(cv[b][i], 0) for i in range(L)] for b in range(G)]            # every lane reads, stale contexts included
            for b in range(G):
                for i in range(L):
                    if R[b][i]:
                        rs[b][i] = r[b][i]
                        rdone[b][i] = True
                        nb, ni = (b, i + 1) if i + 1 < L else (b + 1, 0)
                        if nb < G:
                            cv[nb][ni] = r[b][i]
                            known[nb][ni] = True
        # the ordered pass
        r2 = [[0] * L for _ in range(G)]
        old = [[None] * L for _ in range(G)]
        for b in range(G):
            for i in range(L):
                cur = H.get(cv[b][i], 0)
                r2[b][i] = cur
                if not fin[b][i] and not P[b][i]:
                    old[b][i] = cur
                    H[cv[b][i]] = hw[b * L + i]
        bad = [(b, i) for b in range(G) for i in range(L) if P[b][i] and not fin[b][i] and r2[b][i] != rs[b][i]]
        if not bad:
            for b in range(G):
                for i in range(L):
                    if not fin[b][i]:
                        rf[b][i] = r2[b][i]
            break
        b0, i0 = bad[0]
        stands = [[(b < b0) or (b == b0 and i <= i0) for i in range(L)] for b in range(G)]
        for b in reversed(range(G)):                                                 # take back: the latest write first
            for i in reversed(range(L)):
                if not P[b][i] and not stands[b][i] and not fin[b][i]:
                    H[cv[b][i]] = old[b][i]
        truth = r2[b0][i0]
        for b in range(G):
            for i in range(L):
                if stands[b][i] and not fin[b][i]:
                    rf[b][i] = r2[b][i]
        fin = stands
        if all(all(row) for row in fin):
            break
        nb, ni = (b0, i0 + 1) if i0 + 1 < L else (b0 + 1, 0)
        for b in range(G):
            for i in range(L):
                known[b][i] = stands[b][i] or K0[b][i] or (b, i) == (nb, ni)
        cv[nb][ni] = truth
    last = rf[G - 1][L - 1] if P[G - 1][L - 1] else h[n - 1]
    return [cv[b][i] for b in range(G) for i in range(L)], last, passes


def team_turn(pred, h, hw, c_in, H, H_stale, c_known, G=2, L=64):
    """One turn of decode_passes.hip::cheetah_walk_team (round 6): the speculative reads happen AHEAD of the turn against `H_stale` (H as some earlier moment left
    it — other waves' turns not yet applied; anything at all would do), with lane 0's context taken from the descriptors if `c_known` and left open otherwise;
    under the token: lane 0 patched with the running context `c_in`, what that sets free read from the real H, the ordered pass, the verification; behind a
    wrong read only its CHAIN is guessed again (the lane that now knows its context, the run of predicted lanes it starts, the lane behind that run) unless the
    chain runs on into the next block.  Returns (contexts, next running context, ordered passes)."""
    n = G * L
    P = [[pred[b * L + i] for i in range(L)] for b in range(G)]
    cv = [[0] * L for _ in range(G)]
    K0 = [[False] * L for _ in range(G)]
    for b in range(G):
        for i in range(L):
            if i == 0:
                K0[b][0] = c_known if b == 0 else (not P[b - 1][L - 1])
                cv[b][0] = (c_in if c_known else 0xdead) if b == 0 else h[(b - 1) * L + L - 1]
            else:
                K0[b][i] = not P[b][i - 1]
                cv[b][i] = h[b * L + i - 1]
    known = [row[:] for row in K0]
    fin = [[False] * L for _ in range(G)]
    rdone = [[False] * L for _ in range(G)]
    rs = [[0] * L for _ in range(G)]
    rf = [[0] * L for _ in range(G)]

    def speculate(table):
        while True:
            R = [[P[b][i] and known[b][i] and not rdone[b][i] for i in range(L)] for b in range(G)]
            if not any(any(r) for r in R):
                return
            r = [[table.get(cv[b][i], 0) for i in range(L)] for b in range(G)]
            for b in range(G):
                for i in range(L):
                    if R[b][i]:
                        rs[b][i] = r[b][i]
                        rdone[b][i] = True
                        nb, ni = (b, i + 1) if i + 1 < L else (b + 1, 0)
                        if nb < G:
                            cv[nb][ni] = r[b][i]
                            known[nb][ni] = True

    speculate(H_stale)                                                                # ahead of the turn
    # ---- the turn ----
    if not c_known:
        cv[0][0] = c_in
        K0[0][0] = known[0][0] = True
        speculate(H)
    passes = 0
    while True:
        passes += 1
        r2 = [[0] * L for _ in range(G)]
        old = [[None] * L for _ in range(G)]
        for b in range(G):
            for i in range(L):
                cur = H.get(cv[b][i], 0)
                r2[b][i] = cur
                if not fin[b][i] and not P[b][i]:
                    old[b][i] = cur
                    H[cv[b][i]] = hw[b * L + i]
        bad = [(b, i) for b in range(G) for i in range(L) if P[b][i] and not fin[b][i] and (r2[b][i] != rs[b][i] or not rdone[b][i])]
        if not bad:
            for b in range(G):
                for i in range(L):
                    if not fin[b][i]:
                        rf[b][i] = r2[b][i]
            break
        b0, i0 = bad[0]
        stands = [[(b < b0) or (b == b0 and i <= i0) for i in range(L)] for b in range(G)]
        for b in reversed(range(G)):
            for i in reversed(range(L)):
                if not P[b][i] and not stands[b][i] and not fin[b][i]:
                    H[cv[b][i]] = old[b][i]
        truth = r2[b0][i0]
        for b in range(G):
            for i in range(L):
                if stands[b][i] and not fin[b][i]:
                    rf[b][i] = r2[b][i]
        fin = stands
        if all(all(row) for row in fin):
            break
        cb, start = (b0 + 1, 0) if i0 == L - 1 else (b0, i0 + 1)
        cv[cb][start] = truth
        end = start
        while end < L and P[cb][end]:
            end += 1                                                                 # the lane behind the run (the chain's last)
        if end <= L - 1:
            for i in range(start, end + 1):
                known[cb][i] = False
                rdone[cb][i] = False
            known[cb][start] = True
        else:                                                                        # the chain runs on into the next block: everything behind the wrong read again
            for b in range(G):
                for i in range(L):
                    known[b][i] = fin[b][i] or K0[b][i] or (b, i) == (cb, start)
                    rdone[b][i] = fin[b][i]
        speculate(H)
    last = rf[G - 1][L - 1] if P[G - 1][L - 1] else h[n - 1]
    return [cv[b][i] for b in range(G) for i in range(L)], last, passes


def random_stream(rnd, n, n_hashes, p_pred, max_run=7):
    pred, h, hw, run = [], [], [], 0
    for _ in range(n):
        p = rnd.random() < p_pred and run < max_run
        run = run + 1 if p else 0
        pred.append(p)
        v = rnd.randrange(n_hashes)
        h.append(0 if p else v)
        hw.append(0 if p else (v if rnd.random() < 0.95 else 0))                     # (kDescZero: a MAP quad that read a never-written 0)
    return pred, h, hw


def check(seed, G, groups=6, L=64, n_hashes=40, p_pred=0.4):
    rnd = random.Random(seed)
    pred, h, hw = random_stream(rnd, G * L * groups, n_hashes, p_pred)
    Hs, Hg = {}, {}
    want, c_end = sequential(pred, h, hw, 0, Hs)
    got, c, total = [], 0, 0
    for g in range(groups):
        sl = slice(g * G * L, (g + 1) * G * L)
        ctx, c, passes = grouped(pred[sl], h[sl], hw[sl], c, Hg, G, L)
        got += ctx
        total += passes
    assert got == want and c == c_end, (seed, G)
    assert {k: v for k, v in Hs.items() if v} == {k: v for k, v in Hg.items() if v}, (seed, G)
    return total / groups


def check_team(seed, G=2, groups=8, L=64, n_hashes=40, p_pred=0.4, lag=3, garbage=0.0):
    """the team walk over `groups` turns: every turn speculates against H as it was `lag` turns ago (with a share `garbage` of its entries replaced by noise)"""
    rnd = random.Random(seed)
    pred, h, hw = random_stream(rnd, G * L * groups, n_hashes, p_pred)
    Hs, Hg = {}, {}
    want, c_end = sequential(pred, h, hw, 0, Hs)
    snaps = [dict()]                                                                  # H before turn g
    got, c, total = [], 0, 0
    for g in range(groups):
        sl = slice(g * G * L, (g + 1) * G * L)
        stale = dict(snaps[max(0, g - lag)])
        for k in list(stale):
            if rnd.random() < garbage:
                stale[k] = rnd.randrange(n_hashes)
        c_known = g == 0 or not pred[g * G * L - 1]
        ctx, c, passes = team_turn(pred[sl], h[sl], hw[sl], c, Hg, stale, c_known, G, L)
        got += ctx
        total += passes
        snaps.append(dict(Hg))
    assert got == want and c == c_end, (seed, G, lag)
    assert {k: v for k, v in Hs.items() if v} == {k: v for k, v in Hg.items() if v}, (seed, G, lag)
    return total / groups


if __name__ == "__main__":
    for G in (1, 2, 4):
        for nh, pp in ((5, 0.5), (40, 0.4), (3000, 0.32), (60000, 0.3)):
            avg = sum(check(s, G, n_hashes=nh, p_pred=pp) for s in range(40)) / 40
            print(f"G={G} hashes={nh} p_pred={pp}: ok, passes per group {avg:.2f}")
    for lag in (0, 1, 3):
        for nh, pp in ((5, 0.5), (40, 0.4), (3000, 0.32), (60000, 0.3)):
            avg = sum(check_team(s, 2, n_hashes=nh, p_pred=pp, lag=lag) for s in range(40)) / 40
            print(f"team, reads {lag} turns stale, hashes={nh} p_pred={pp}: ok, ordered passes per turn {avg:.2f}")
