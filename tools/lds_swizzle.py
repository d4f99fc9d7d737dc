#!/usr/bin/env python3
"""Exhaustive search for GF(2)-linear XOR swizzles of the LDS images of csrc/mpconv_bwd_sg.hip that make EVERY structured
access pattern bank-conflict-free at once, under gfx950's per-instruction lane groups (MI355X_MICROARCH.md, LDS):
ds_read_b128 = four groups of 16 NON-contiguous lanes {0-3,12-15,20-27}, ...; ds_read_b64 = two halves of 32 lanes; a bank row
is 256 bytes = 16 slots of 16 bytes.  Lane (li = lane & 15, lk = lane >> 4).

 x image, rows of 8 slots (two rows per bank row): physical row = row with bit 0 replaced by parity(h & row), slot ^= g(row).
   G  projection operand reads (b128): row = li, slots lk and lk + 4
   H  dW transposed reads (b64): rows 8 lk + j (j = 0..7), slot li >> 1, half li & 1
 P / dP image, rows of 32 slots: slot ^= g4(row)
   D  dx operand reads (b128): row = li, slot lk + 4 ks;  E  dW transposed reads (b64);  A  P stores (b64, 2-way either way)
Prints the padded-stride costs (what the kernel used before) and the swizzles that cost nothing."""
import itertools

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
par = lambda x: bin(x).count('1') & 1


def gf(M, row):
    return sum(par(m & row) << i for i, m in enumerate(M))


def padded_cost(a):
    """ds_read_b128 cycles per instruction (4 = conflict-free) for slot = a * li + lk."""
    tot = 0
    for g in G128:
        seen = {}
        for lane in g:
            s = (a * (lane & 15) + (lane >> 4)) % 16
            seen[s] = seen.get(s, 0) + 1
        tot += max(seen.values())
    return tot


def cost_x(h, M):
    c = 0
    for extra in (0, 4):
        for g in G128:
            seen = {}
            for lane in g:
                li, lk = lane & 15, lane >> 4
                pos = par(h & li) * 8 + (((lk + extra) & 7) ^ gf(M, li))
                seen[pos] = seen.get(pos, 0) + 1
            c += max(seen.values()) - 1
    for j in range(8):
        for base in (0, 2):
            seen = {}
            for li in range(16):
                for lk in (base, base + 1):
                    row = (8 * lk + j) & 15
                    pos = (par(h & row) * 8 + ((li >> 1) ^ gf(M, row))) * 2 + (li & 1)
                    seen[pos] = seen.get(pos, 0) + 1
            c += max(seen.values()) - 1
    return c


def cost_p(M):
    c = 0
    for ks in range(8):
        for g in G128:
            seen = {}
            for lane in g:
                li, lk = lane & 15, lane >> 4
                pos = ((lk + 4 * ks) & 15) ^ gf(M, li)
                seen[pos] = seen.get(pos, 0) + 1
            c += max(seen.values()) - 1
    for wq in range(4):
        for j in range(8):
            for base in (0, 2):
                seen = {}
                for li in range(16):
                    for lk in (base, base + 1):
                        row = (8 * lk + j) & 15
                        pos = (((8 * wq + (li >> 1)) & 15) ^ gf(M, row)) * 2 + (li & 1)
                        seen[pos] = seen.get(pos, 0) + 1
                c += max(seen.values()) - 1
    return c


if __name__ == '__main__':
    print('padded strides, ds_read_b128 cycles per operand read (4 = conflict-free):',
          {16 * a: padded_cost(a) for a in (9, 10, 17, 18, 33, 34)})
    free = [(h, M) for h in (9, 11, 13, 15) for M in itertools.product(range(16), repeat=3) if cost_x(h, M) == 0]
    print('x image: %d conflict-free (h, g) pairs; used: h = 9 (bit0 ^ bit3), g = (row >> 1) & 7 ->' % len(free), cost_x(9, (2, 4, 8)))
    print('P / dP image: g = row & 15 ->', cost_p((1, 2, 4, 8)), '(g = 0, i.e. unpadded and unswizzled: %d)' % cost_p((0, 0, 0, 0)))
