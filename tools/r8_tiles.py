#!/usr/bin/env python3
"""Position tiles of the fused res8 kernel's channel-block-major map (res8_f16x3.hip) + their LDS bank model.

Map: [part 2][channel block 6][cell 384] x 16 B; cell(y, x) = 14 (y + 1) + x + 1.  A ds_read_b128 of a B fragment is served in four
16-lane groups, each holding all 16 positions of the tile (8 lanes of one k-group + the complementary 8 of the next, whose plane
starts 0 mod 256 B further on): it is conflict-free iff the tile's 16 cells are distinct mod 16.  The 325 positions fall into residue
classes of 18..22 cells, so tiles 0..19 take one position per class (a class that has run out leaves a pad lane, which clones lane
0 of its tile: same address = broadcast) and tile 20 takes the 15 left over (2-way conflicts there only).

Prints the table for res8_f16x3.hip (R8H_POS: position | 0x8000 for pad lanes) and the modelled LDS cycles per layer."""
import sys
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from lds_bank_model import read_b128, write_b64

H, W, RS = 25, 13, 14


def cell(p):
    return (p // W + 1) * RS + p % W + 1


def build():
    by_res = {r: [] for r in range(16)}
    for p in range(H * W):
        by_res[cell(p) % 16].append(p)
    tiles = []
    for t in range(20):
        row = [by_res[r][t] if t < len(by_res[r]) else None for r in range(16)]
        tiles.append(row)
    rest = [p for r in range(16) for p in by_res[r][20:]]
    assert len(rest) <= 16
    tiles.append(rest + [None] * (16 - len(rest)))
    out = []
    for row in tiles:
        first = next(p for p in row if p is not None)
        out.append([(p, 0) if p is not None else (first, 1) for p in row])
    return out


def model(tiles):
    rd = wr = 0
    plane = 384 * 16
    for row in tiles:
        cells = [cell(p) for p, _ in row]
        for s in range(14):
            addr = []
            for l in range(64):
                bi = min(4 * s + (l >> 4), 53)
                tap, cb = divmod(bi, 6)
                ty, tx = divmod(tap, 3)
                addr.append((cb * plane + (cells[l & 15] + (ty - 1) * RS + (tx - 1)) * 16) // 4)
            rd += 2 * read_b128(addr)
        for m in range(3):
            addr = [((2 * m + (l >> 5)) * plane + cells[l & 15] * 16 + 8 * ((l >> 4) & 1)) // 4 for l in range(64)]
            wr += 2 * write_b64(addr)
    return rd, 21 * 14 * 2 * 4, wr, 21 * 3 * 2 * 4


if __name__ == "__main__":
    tiles = build()
    seen = sorted(p for row in tiles for p, pad in row if not pad)
    assert seen == list(range(H * W)), "every position exactly once"
    print("LDS cycles per layer (read, ideal, write, ideal):", model(tiles))
    print("pads per tile:", [sum(pad for _, pad in row) for row in tiles])
    vals = [p | (0x8000 if pad else 0) for row in tiles for p, pad in row]
    print("static const unsigned short R8H_POS_HOST[21 * 16] = {")
    for i in range(0, len(vals), 16):
        print("    " + ", ".join(f"0x{v:04x}" for v in vals[i:i + 16]) + ",")
    print("};")
