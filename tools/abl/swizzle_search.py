#!/usr/bin/env python3
"""Brute-force search of the XOR swizzle keys of the direct 3x3 convolution's LDS tiles (csrc/conv.hip, namespace dconv; CPU only).

Model of a `ds_read_b128` (MI355X_MICROARCH.md, LDS table): 64 banks x 4 B, bank = (a / 4) mod 64; the 64 lanes are served in four
NON-contiguous groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} --, each
lane taking four consecutive banks, i.e. one of 16 "chunk columns" (byte address / 16 mod 16).  A group is conflict-free when its 16
lanes hit 16 different chunk columns; its cost is the largest number of DIFFERENT addresses on one column.

A fragments (input tile, [pixel][CI channels], PB = 2 CI bytes per pixel): lane (g, li) of row block mi of wave w reads pixel
P = (y + dy) 32 + x + dx with (y, x) = divmod(64 w + 16 mi + li, 30), chunk c(g, k-step) of that pixel, stored at position c ^ key(P).
B fragments (filter bank, k-step tiles [CO rows][64 bytes]): lane (g, li) reads row n = (li >> 2) (CO / 4) + 4 ni + (li & 3), chunk g,
stored at g ^ key(n).

The search tries key(P) = (P >> s) & m for shifts 0-4 and masks up to the pixel's chunk count and prints the worst phase cost over
all taps / k-steps / row blocks; cost 1 everywhere = conflict-free.  The shipped keys are the best of the family: filter bank 1
(conflict-free), input tile 2 (a 16-pixel fragment that wraps from one 30-pixel tile row to the next skips two LDS pixels: two lanes
then share a column; without the swizzle 4 / 3 / 2 for CI = 64 / 32 / 16).
"""
import itertools


def _r(*spans):
    return [l for a, b in spans for l in range(a, b + 1)]


GROUPS = [_r((0, 3), (12, 15), (20, 27)), _r((4, 11), (16, 19), (28, 31)), _r((32, 35), (44, 47), (52, 59)), _r((36, 43), (48, 51), (60, 63))]


def phase_cost(addrs):
    """addrs: 64 byte addresses of one b128 read -> worst lanes-per-chunk-column over the four 16-lane phases"""
    worst = 0
    for grp in GROUPS:
        cols = {}
        for lane in grp:
            a = addrs[lane]
            cols.setdefault((a >> 4) & 15, set()).add(a)       # the same address twice is a broadcast, not a conflict
        worst = max(worst, max(len(v) for v in cols.values()))
    return worst


def a_cost(CI, key):
    PB, NC = 2 * CI, CI // 8
    worst = 0
    for wave, mi in itertools.product(range(4), range(4)):
        for dy, dx in itertools.product(range(3), range(3)):
            for half in range(max(1, CI // 32)):           # CI = 64: two k-steps per tap (channels 0-31 / 32-63)
                addrs = []
                for lane in range(64):
                    g, li = lane >> 4, lane & 15
                    q = min(64 * wave + 16 * mi + li, 239)
                    y, x = divmod(q, 30)
                    if CI == 16:                           # a k-step = two taps x 16 channels: lane groups 0-1 first tap, 2-3 second
                        t = min(3 * dy + dx + (g >> 1), 8)
                        P = (y + t // 3) * 32 + x + t % 3
                        chunk = g & 1
                    else:
                        P = (y + dy) * 32 + x + dx
                        chunk = 4 * half + g if CI == 64 else g
                    addrs.append(P * PB + ((chunk ^ (key(P) & (NC - 1))) << 4))
                worst = max(worst, phase_cost(addrs))
    return worst


def b_cost(CO, key):
    worst = 0
    for ni in range(CO // 16):
        addrs = []
        for lane in range(64):
            g, li = lane >> 4, lane & 15
            n = (li >> 2) * (CO // 4) + ni * 4 + (li & 3)
            addrs.append(n * 64 + ((g ^ (key(n) & 3)) << 4))
        worst = max(worst, phase_cost(addrs))
    return worst


def main():
    print("A fragments (input tile): worst lanes per chunk column in a 16-lane phase, key(P) = (P >> shift) & mask")
    for CI in (64, 32, 16):
        NC = CI // 8
        rows = []
        for shift, mask in itertools.product(range(5), [m for m in (0, 1, 3, 7) if m < NC]):
            rows.append((a_cost(CI, lambda P, s=shift, m=mask: (P >> s) & m), shift, mask))
        rows.sort()
        print(f"  CI = {CI:2d}: best " + ", ".join(f"(P >> {s}) & {m}: {c}" for c, s, m in rows[:4]) + f"   | no swizzle: {a_cost(CI, lambda P: 0)}")
    shipped = {64: lambda P: P & 7, 32: lambda P: (P >> 1) & 3, 16: lambda P: 0}
    print("  shipped akey: " + ", ".join(f"CI = {ci}: {a_cost(ci, k)}" for ci, k in shipped.items()))
    print("B fragments (filter bank, 64-byte rows): key(n) = (n >> shift) & 3")
    for CO in (64, 32, 16):
        print(f"  CO = {CO:2d}: " + ", ".join(f"shift {s}: {b_cost(CO, lambda n, s=s: (n >> s) & 3)}" for s in range(6)) + f"   | no swizzle: {b_cost(CO, lambda n: 0)}")
    shipped_b = {64: 3, 32: 2, 16: 1}
    print("  shipped bkey: " + ", ".join(f"CO = {co}: {b_cost(co, lambda n, s=s: (n >> s) & 3)}" for co, s in shipped_b.items()))


if __name__ == "__main__":
    main()
