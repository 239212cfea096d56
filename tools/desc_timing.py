"""Debug: phase cycle counters of describe_one (library built with -DVFSMS_DESC_TIMING)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagestitch_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("VFSMS_TIMING_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvfsms_timing.so"))
import imagestitch_amd as isa
from imagestitch_amd.synthetic import SyntheticGrid
eng = isa.Engine(0)
g = SyntheticGrid(10, 9, 2048)
N = 8
tiles = g.tiles(range(N + 1)); hs = [eng.tile_upload(t) for t in tiles]
ra = isa.roi_rect(tiles[0].shape, 1, "first", 0.2); rb = isa.roi_rect(tiles[0].shape, 1, "second", 0.2)
jobs = [(hs[k], hs[k + 1], ra[0], ra[1], rb[0], rb[1], ra[2], ra[3]) for k in range(N)]
rows = eng.attempt_surf_batch(jobs); eng.set_keypoint_capacity(int(rows[:, 4:6].max() * 1.5) + 1024)
out0 = np.zeros(8, np.uint64); eng.lib.vfsms_debug_desc_cycles(out0.ctypes.data_as(ctypes.c_void_p))
rows = eng.attempt_surf_batch(jobs)
out = np.zeros(8, np.uint64); eng.lib.vfsms_debug_desc_cycles(out.ctypes.data_as(ctypes.c_void_p))
d = (out - out0).astype(np.float64)
nk = rows[:, 4:6].sum()
names = ["prologue", "stageA", "outputsA", "stageB", "outputsB", "descriptor", "ticket"]
print("keypoints", nk, "total block-cycles %.3g (clock64 ticks)" % d.sum())
for n, v in zip(names, d[:7]):
    print("  %-10s %6.1f %%   %.0f ticks/keypoint" % (n, 100 * v / d.sum(), v / nk))

tr = np.zeros(4, np.uint64); eng.lib.vfsms_debug_desc_trips(tr.ctypes.data_as(ctypes.c_void_p))
print("wave trips (both runs): interior strips %d (x4 samples/lane), border strips all-in %d (x%d), border per-sample %d" % (tr[0], tr[1], 1, tr[2]))

uc = np.zeros(4, np.uint64); eng.lib.vfsms_debug_desc_unit_cycles(uc.ctypes.data_as(ctypes.c_void_p))
print("units (both runs): interior %d, border %d; wave-cycles per unit: interior %.0f, border %.0f; wave-cycles in stage_rows %.4g (in units %.4g)" % (
    tr[0], tr[2], uc[0] / max(float(tr[0]), 1), uc[1] / max(float(tr[2]), 1), float(uc[2]), float(uc[0] + uc[1])))

# BF filter sweeps: per-wave cycles by phase (both runs)
bf = np.zeros(16, np.uint64)
if hasattr(eng.lib, "vfsms_debug_bf_cycles") and eng.lib.vfsms_debug_bf_cycles(bf.ctypes.data_as(ctypes.c_void_p)) == 0:
    bf = bf.reshape(2, 8).astype(np.float64)
    for p in (0, 1):
        tot, waves = bf[p, 5], max(bf[p, 6], 1)
        print("bf pass %d: %d waves, %.0f cycles per wave; prologue %.1f %%, barrier wait %.1f %%, put+fetch %.1f %%, lds+mfma+min %.1f %%, append %.1f %%, rest %.1f %%" % (
            p, waves, tot / waves, *[100 * bf[p, q] / tot for q in range(5)], 100 * (tot - bf[p, :5].sum()) / tot))
