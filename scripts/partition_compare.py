"""Rectangles of blocks (what sharded.py cuts) against Morton-ordered block ranges (north_star's wording), on the block
grid of a uniform scene: border blocks per rank (their particles are some neighbour's ghosts, every step) and the
number of neighbouring ranks.  python scripts/partition_compare.py BX BY RANKS"""
import sys
import numpy as np

def split(n):
    x = n & 0xFFFF
    x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F
    x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555
    return x

def stats(owner, ranks):
    by, bx = owner.shape
    pad = np.full((by + 2, bx + 2), -1, np.int64); pad[1:-1, 1:-1] = owner
    border = np.zeros(ranks, np.int64); neigh = [set() for _ in range(ranks)]
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            if dx == 1 and dy == 1: continue
            other = pad[dy:dy + by, dx:dx + bx]
            diff = (other != owner) & (other >= 0)
            for r in range(ranks):
                m = diff & (owner == r)
                neigh[r].update(np.unique(other[m]).tolist())
    anyd = np.zeros_like(owner, bool)
    for dy in (0, 1, 2):
        for dx in (0, 1, 2):
            other = pad[dy:dy + by, dx:dx + bx]
            anyd |= (other != owner) & (other >= 0)
    for r in range(ranks):
        border[r] = int((anyd & (owner == r)).sum())
    return border, [len(s) for s in neigh]

bx, by, ranks = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
px = int(np.floor(np.sqrt(ranks)))
while ranks % px: px -= 1
py = ranks // px
xs = np.arange(bx) * px // bx; ys = np.arange(by) * py // by
rect = (ys[:, None] * px + xs[None, :]).astype(np.int64)
yy, xx = np.mgrid[0:by, 0:bx]
key = np.vectorize(split)(xx) | (np.vectorize(split)(yy) << 1)
order = np.argsort(key.reshape(-1), kind="stable")
mort = np.empty(bx * by, np.int64); mort[order] = np.arange(bx * by) * ranks // (bx * by)
mort = mort.reshape(by, bx)
for name, own in (("rectangles %dx%d" % (px, py), rect), ("Morton ranges", mort)):
    b, n = stats(own, ranks)
    print("%-18s blocks/rank %s  border blocks/rank max %d mean %.0f (%.2f %% of a rank)  neighbours/rank max %d mean %.1f" %
          (name, np.bincount(own.reshape(-1), minlength=ranks).tolist()[:4], b.max(), b.mean(), 100.0 * b.mean() * ranks / (bx * by),
           max(n), float(np.mean(n))))
