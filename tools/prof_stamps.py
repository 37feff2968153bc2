"""Offline look at a DENSITY_HIP_PROF_DUMP (rotor.hip: rot_prof_report): hop / iteration / lateness statistics per wave.
usage: python tools/prof_stamps.py <prefix> [decode_waves decode_rounds_of]"""
import sys
import numpy as np
prefix = sys.argv[1]
for what, W in (("decode", int(sys.argv[2]) if len(sys.argv) > 2 else 12), ("encode", 8)):
    h = np.fromfile(f"{prefix}.{what}.bin", dtype=np.uint64).astype(np.int64)
    ts = h[128:128 + 4 * 2048].reshape(2048, 4)
    n = np.flatnonzero((ts[:, 1] > 0) & (ts[:, 2] > 0)).max() + 1
    arrive, seen, done, end = ts[:n, 0], ts[:n, 1], ts[:n, 2], ts[:n, 3]
    hop = np.diff(seen)[17:]
    r = np.arange(W + 17, n)
    it, tail, head, late = arrive[r] - done[r - W], end[r - W] - done[r - W], arrive[r] - end[r - W], arrive[r] - done[r - 1]
    crit = (done - seen)[17:]
    hp = seen[r] - seen[r - 1]
    print(f"{what}: {n} rounds, hop mean {hop.mean():.0f} median {np.median(hop):.0f}; critical section mean {crit.mean():.0f} p90 {np.percentile(crit, 90):.0f}")
    print(f"  a wave's iteration (exchanges done -> next arrival): mean {it.mean():.0f} p50 {np.percentile(it, 50):.0f} p90 {np.percentile(it, 90):.0f} p99 {np.percentile(it, 99):.0f}"
          f" = to the end of the round {tail.mean():.0f} + on to the arrival {head.mean():.0f}")
    print(f"  late arrivals {(late > 0).sum()} of {len(r)}, mean lateness {late[late > 0].mean() if (late > 0).any() else 0:.0f}; hop when on time {hp[late <= 0].mean():.0f}, when late {hp[late > 0].mean() if (late > 0).any() else 0:.0f}")
    print("  per wave (late arrivals / mean iteration):", "  ".join(f"w{w}: {((arrive[r[r % W == w]] - done[r[r % W == w] - 1]) > 0).sum()}/{(arrive[r[r % W == w]] - done[r[r % W == w] - W]).mean():.0f}" for w in range(W)))
