"""Dev aid: where the EPI_RESID24 / EPI_LNF epilogues differ from the numpy restatement (rows, columns, tiles)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from image_search_amd import ops
import test_lnfold_gpu as T

F32 = np.float32
grid = sys.argv[1] if len(sys.argv) > 1 else "3"
os.environ["MI_OP_GRID"] = grid
for (m, n, k) in ((1000, 256, 128), (2304, 512, 256), (700, 1024, 320)):
    rng = np.random.default_rng(11)
    x = rng.integers(-2, 3, (m, k)).astype(F32)
    w = rng.integers(-1, 2, (n, k)).astype(F32)
    b = rng.integers(-3, 4, n).astype(F32)
    res = (rng.integers(-4000, 4001, (m, n)) / 16).astype(F32)
    want = (res + (x @ w.T + b)).astype(F32)
    got, hi, part, stats = ops.linear_resid24(x, w, b, res)
    bad = got != want
    print(f"grid {grid} shape {(m, n, k)}: {bad.sum()} of {bad.size} elements differ")
    if bad.any():
        r, c = np.nonzero(bad)
        print("  rows:", np.unique(r)[:40], "... count", len(np.unique(r)))
        print("  cols:", np.unique(c)[:40], "... count", len(np.unique(c)))
        print("  row tiles:", np.unique(r // 256), "col tiles:", np.unique(c // 256))
        print("  row%16:", np.unique(r % 16), "col%64:", np.unique(c % 64)[:64])
        i = 0
        print("  first:", r[i], c[i], "got", got[r[i], c[i]], "want", want[r[i], c[i]], "res", res[r[i], c[i]], "delta", want[r[i], c[i]] - res[r[i], c[i]])
        d = got - res
        dd = (x @ w.T + b)
        print("  got-res vs delta at first 5:", [(float(d[r[j], c[j]]), float(dd[r[j], c[j]])) for j in range(min(5, len(r)))])
    pw = T.block_sums(want)
    pb = part.view(np.uint32) != pw.view(np.uint32)
    print("  part mismatches:", pb.sum(), "of", pb.size)
    if pb.any():
        r, bl, c = np.nonzero(pb)
        print("   rows", np.unique(r)[:20], "blocks", np.unique(bl), "comp", np.unique(c))
