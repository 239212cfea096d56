"""Stress check of the fuse's statistics hand-off (round 6: no agent fences): the 90-tile mosaic assembled REPS times, every result hashed; with a
second process keeping the GPU busy (tools/microbench.py in the background) the workgroups of a statistics launch arrive unevenly.  All hashes
must be one value -- and equal to the reference build's (VFSMS_LIB=... python tools/fuse_stress.py 3 prints its hash).
   python tools/fuse_stress.py [reps] [rows cols tile]"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagestitch_amd as isa
from imagestitch_amd.synthetic import SyntheticGrid

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rows, cols, tile = (int(a) for a in (sys.argv[2:5] if len(sys.argv) >= 5 else (10, 9, 1024)))
eng = isa.Engine(0)
g = SyntheticGrid(rows, cols, tile)
tiles = g.tiles(range(g.n_tiles), threads=8)
hs = [eng.tile_upload(t) for t in tiles]
offs = [[0, 0]] + [list(map(int, o)) for o in g.true_offsets()]
shapes = [(g.th, g.tw)] * g.n_tiles
offsetList, rangeX, rangeY, R, C = isa.Stitcher._layout(shapes, offs)
rois = [(max(offsetList[i][0], rangeX[i - 1][0]), max(offsetList[i][1], rangeY[i - 1][0]), min(offsetList[i][0] + g.th, rangeX[i - 1][1]),
         min(offsetList[i][1] + g.tw, rangeY[i - 1][1])) for i in range(1, g.n_tiles)]
geom = [(offsetList[0][0], offsetList[0][1], 0, 0, 0, 0, 0, 0, -1)] + [(offsetList[i][0], offsetList[i][1]) + tuple(rois[i - 1]) + (offs[i][0], offs[i][1], 0)
                                                                        for i in range(1, g.n_tiles)]
seen = {}
t0 = time.perf_counter()
for r in range(reps):
    cv = eng.canvas_create(R, C, 1)
    eng.canvas_assemble_resident(cv, hs, geom)
    out = eng.canvas_download(cv, R, C, 1)
    eng.canvas_free(cv)
    h = hashlib.sha256(out.tobytes()).hexdigest()[:16]
    seen[h] = seen.get(h, 0) + 1
print("fuse_stress reps=%d grid=%dx%d tile=%d lib=%s: %d distinct result(s) %s  (%.1f s)" % (reps, rows, cols, tile, os.environ.get("VFSMS_LIB", "in-tree"), len(seen), seen, time.perf_counter() - t0))
