"""Synthetic micrograph grids with exact integer ground-truth offsets (SURVEY section 8d "Synthetic inputs").

The texture is a pure function of GLOBAL canvas coordinates (hashed-lattice value noise over several
octaves plus one Gaussian dot or short line segment per 400 px^2 cell), so any tile of any grid size can be produced on its own -- per rank, on the fly, without ever
materialising the canvas (a 32x32 grid of 4096^2 tiles would be 16 GB).  Tiles follow the column-major
serpentine shooting path of the reference's dendriticCrystal demo set (down, step right, up, ...), with a
nominal 10 % overlap, integer jitter U{-8..8} on both axes, per-tile gain U(0.97, 1.03) and additive
Gaussian noise sigma = 2, so overlapping pixels are similar but never identical.
"""
import numpy as np

DEFAULT_SEED = 20190158


def _hash_u32(ix, iy, salt):
    """Vectorised 2-D integer hash -> uint32 (splitmix-style avalanche)."""
    x = (ix.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (iy.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ np.uint64(salt)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def _octave(gy0, gx0, h, w, cell, salt):
    """Smooth-interpolated lattice noise in [0, 1) over the window [gy0, gy0+h) x [gx0, gx0+w)."""
    gy = np.arange(gy0, gy0 + h, dtype=np.int64)
    gx = np.arange(gx0, gx0 + w, dtype=np.int64)
    cy0, cx0 = gy[0] // cell, gx[0] // cell
    ny, nx = int(gy[-1] // cell - cy0 + 2), int(gx[-1] // cell - cx0 + 2)
    ly, lx = np.meshgrid(np.arange(cy0, cy0 + ny), np.arange(cx0, cx0 + nx), indexing="ij")
    with np.errstate(over="ignore"):
        L = _hash_u32(lx + (1 << 20), ly + (1 << 20), salt).astype(np.float32) * np.float32(1.0 / 4294967296.0)
    iy = (gy // cell - cy0).astype(np.intp); ix = (gx // cell - cx0).astype(np.intp)
    fy = ((gy % cell).astype(np.float32) / np.float32(cell)); fx = ((gx % cell).astype(np.float32) / np.float32(cell))
    fy = fy * fy * (3 - 2 * fy); fx = fx * fx * (3 - 2 * fx)
    top = L[iy][:, ix] * (1 - fx)[None, :] + L[iy][:, ix + 1] * fx[None, :]
    bot = L[iy + 1][:, ix] * (1 - fx)[None, :] + L[iy + 1][:, ix + 1] * fx[None, :]
    return top * (1 - fy)[:, None] + bot * fy[:, None]


_OCTAVES = ((192, 0.9), (48, 0.8), (14, 1.0), (7, 1.1), (4, 0.9))


_BLOB_CELL = 20          # one blob per 20 x 20 px cell = one per 400 px^2
_BLOB_REACH = 48         # no blob reaches further than this from its cell
_BLOB_GAIN = 1.1         # blob layer amplitude relative to the unit-variance noise
_BLOB_NORM = 0.898       # std of (noise + _BLOB_GAIN x blob layer), measured once (noise alone: 0.734): the sum gets unit variance


def _gauss_kernel(sigma):
    r = int(np.ceil(3 * sigma))
    y, x = np.mgrid[-r:r + 1, -r:r + 1].astype(np.float32)
    return np.exp(-(x * x + y * y) / np.float32(2 * sigma * sigma)).astype(np.float32)


_KERNELS = {}


def blob_window(gy0, gx0, h, w, seed=DEFAULT_SEED):
    """The dendrite-like layer of SURVEY 8d: per 20 x 20 px cell of the infinite canvas one feature, a pure function of the cell's
    GLOBAL index -- a Gaussian dot of radius 2..12 px (sigma = radius / 2.5) or a short line segment (a chain of narrow dots along
    a random direction), bright or dark.  Gives SURF / FAST the blob and ridge structure micrographs have (value noise alone yields
    a third fewer keypoints than the real dendriticCrystal strips)."""
    acc = np.zeros((h + 2 * _BLOB_REACH, w + 2 * _BLOB_REACH), np.float32)
    cy0, cy1 = (gy0 - _BLOB_REACH) // _BLOB_CELL, (gy0 + h + _BLOB_REACH) // _BLOB_CELL
    cx0, cx1 = (gx0 - _BLOB_REACH) // _BLOB_CELL, (gx0 + w + _BLOB_REACH) // _BLOB_CELL
    cy, cx = np.meshgrid(np.arange(cy0, cy1 + 1), np.arange(cx0, cx1 + 1), indexing="ij")
    cy, cx = cy.ravel(), cx.ravel()
    with np.errstate(over="ignore"):
        hsh = [_hash_u32(cx + (1 << 20), cy + (1 << 20), seed * 977 + 31 * k) for k in range(6)]
    u = [hh.astype(np.float64) / 4294967296.0 for hh in hsh]
    py = cy * _BLOB_CELL + (u[0] * _BLOB_CELL).astype(np.int64) - gy0 + _BLOB_REACH       # feature centre in the padded window
    px = cx * _BLOB_CELL + (u[1] * _BLOB_CELL).astype(np.int64) - gx0 + _BLOB_REACH
    radius = 2 + (u[2] * 11).astype(np.int64)                                              # 2 .. 12
    sign = np.where(u[3] < 0.5, -1.0, 1.0).astype(np.float32)
    is_line = u[4] < 0.5
    theta = u[5] * np.pi
    H, W = acc.shape

    def stamp(y, x, key, amp):
        k = _KERNELS.get(key)
        if k is None:
            k = _KERNELS[key] = _gauss_kernel(key / 10.0)
        r = k.shape[0] // 2
        y0, y1, x0, x1 = y - r, y + r + 1, x - r, x + r + 1
        if y1 <= 0 or x1 <= 0 or y0 >= H or x0 >= W:
            return
        ky0, kx0 = max(0, -y0), max(0, -x0)
        ky1, kx1 = k.shape[0] - max(0, y1 - H), k.shape[1] - max(0, x1 - W)
        acc[max(y0, 0):min(y1, H), max(x0, 0):min(x1, W)] += amp * k[ky0:ky1, kx0:kx1]
    for n in range(len(cy)):
        if is_line[n]:
            L = 6 + 2 * int(radius[n])                           # 10 .. 30 px long, sigma 1.4
            dy, dx = np.sin(theta[n]), np.cos(theta[n])
            for t in range(-L // 2, L // 2 + 1, 2):
                stamp(int(py[n] + round(t * dy)), int(px[n] + round(t * dx)), 14, sign[n] * np.float32(0.8))
        else:
            stamp(int(py[n]), int(px[n]), int(radius[n]) * 4, sign[n] * np.float32(1.6))     # sigma = radius / 2.5
    return acc[_BLOB_REACH:_BLOB_REACH + h, _BLOB_REACH:_BLOB_REACH + w]


def texture_window(gy0, gx0, h, w, seed=DEFAULT_SEED, blobs=True):
    """float32 texture (roughly mean 0, unit variance) over a window of the infinite canvas: (a) multi-octave value noise +
    (b) the blob / line-segment layer of SURVEY 8d (blobs=False gives round 1's noise-only texture)."""
    acc = np.zeros((h, w), np.float32)
    tot = 0.0
    for k, (cell, amp) in enumerate(_OCTAVES):
        acc += np.float32(amp) * (_octave(gy0, gx0, h, w, cell, seed * 131 + k) - np.float32(0.5))
        tot += amp * amp / 12.0
    acc /= np.float32(np.sqrt(tot))
    if not blobs:
        return acc
    return (acc + np.float32(_BLOB_GAIN) * blob_window(gy0, gx0, h, w, seed)) / np.float32(_BLOB_NORM)


class SyntheticGrid:
    """rows x cols tiles of tile_h x tile_w on a column-major serpentine path."""

    def __init__(self, rows, cols, tile_h, tile_w=None, overlap=0.10, jitter=8, seed=DEFAULT_SEED, blobs=True):
        self.blobs = bool(blobs)
        self.rows, self.cols = int(rows), int(cols)
        self.th, self.tw = int(tile_h), int(tile_w or tile_h)
        self.seed = int(seed)
        rng = np.random.default_rng(self.seed)
        step_y = self.th - int(round(overlap * self.th))
        step_x = self.tw - int(round(overlap * self.tw))
        jy = rng.integers(-jitter, jitter + 1, (self.rows, self.cols))
        jx = rng.integers(-jitter, jitter + 1, (self.rows, self.cols))
        self.path = []
        for c in range(self.cols):
            rr = range(self.rows) if c % 2 == 0 else range(self.rows - 1, -1, -1)
            for r in rr:
                self.path.append((r, c))
        origins = np.zeros((self.rows, self.cols, 2), np.int64)
        for r in range(self.rows):
            for c in range(self.cols):
                origins[r, c] = (r * step_y + jy[r, c] + 64, c * step_x + jx[r, c] + 64)
        self.origins = origins
        self.n_tiles = len(self.path)
        self.n_pairs = self.n_tiles - 1

    def origin(self, k):
        r, c = self.path[k]
        return int(self.origins[r, c, 0]), int(self.origins[r, c, 1])

    def true_offsets(self):
        """[[dx, dy]] for consecutive path tiles: row / column shift of tile k+1's origin in tile k's frame
        (the reference's offset convention, Stitcher.py:26-27)."""
        out = []
        for k in range(self.n_pairs):
            (y0, x0), (y1, x1) = self.origin(k), self.origin(k + 1)
            out.append([y1 - y0, x1 - x0])
        return out

    def true_directions(self):
        out = []
        for dx, dy in self.true_offsets():
            out.append(1 if dx > self.th // 2 else 3 if dx < -self.th // 2 else 2 if dy > 0 else 4)
        return out

    def tile(self, k):
        """uint8 tile k of the path (mean 128, sigma ~45, per-tile gain and noise)."""
        y0, x0 = self.origin(k)
        t = texture_window(y0, x0, self.th, self.tw, self.seed, self.blobs)
        rng = np.random.default_rng(1000 + k + self.seed % 1000)
        gain = rng.uniform(0.97, 1.03)
        img = 128.0 + 45.0 * gain * t + rng.normal(0.0, 2.0, t.shape).astype(np.float32)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def tiles(self, ks=None, threads=8, processes=0):
        """tiles ks (default: all) in order.  threads: a thread pool (the noise octaves release the interpreter lock, the blob layer's stamping
        loop does not: ~2x at best).  processes > 1: worker PROCESSES instead (spawned, so that a parent that holds a GPU context is not
        forked) -- the 1024 tiles of 4096 x 4096 of BASELINE configs[4] are 100 core-minutes of stamping; an iterator in this case (a tile is
        handed over and dropped: 17 GB never sit in one list)."""
        ks = list(range(self.n_tiles)) if ks is None else list(ks)
        if processes and processes > 1 and len(ks) > 1:
            return self._tiles_by_processes(ks, processes)
        if threads <= 1 or len(ks) < 2:
            return [self.tile(k) for k in ks]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            return list(ex.map(self.tile, ks))

    def _tiles_by_processes(self, ks, processes):
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        args = (self.rows, self.cols, self.th, self.tw, self.seed, self.blobs, self.origins)
        with ctx.Pool(min(processes, len(ks)), initializer=_worker_init, initargs=(args,)) as pool:
            for t in pool.imap(_worker_tile, ks, chunksize=1):
                yield t


_WORKER_GRID = None


def _worker_init(args):
    global _WORKER_GRID
    rows, cols, th, tw, seed, blobs, origins = args
    g = SyntheticGrid.__new__(SyntheticGrid)
    g.rows, g.cols, g.th, g.tw, g.seed, g.blobs, g.origins = rows, cols, th, tw, seed, blobs, origins
    g.path = []
    for c in range(cols):
        g.path += [(r, c) for r in (range(rows) if c % 2 == 0 else range(rows - 1, -1, -1))]
    g.n_tiles = len(g.path); g.n_pairs = g.n_tiles - 1
    _WORKER_GRID = g


def _worker_tile(k):
    return _WORKER_GRID.tile(k)


def line_scan(n=4, h=1024, w=1280, bar=48):
    """A zircon-like line scan (Main.py:29-51): n tiles of h x w, tile k+1 to the LEFT of tile k (direction 4, full-image search), with a
    static data bar burned into the bottom rows of every tile (pixel-identical between tiles, like zirconCL's) -> (tiles, true offsets)."""
    rng = np.random.default_rng(77)
    step = w - int(0.2 * w)
    bar_px = rng.integers(0, 256, (bar, w), dtype=np.uint8)
    bar_px[:, ::7] = 255
    tiles, offs = [], []
    x = 5000 + max(0, n - 4) * step
    for k in range(n):
        jy, jx = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
        y0, x0 = 300 + jy, x - k * step + jx
        t = texture_window(y0, x0, h, w)
        img = np.clip(np.rint(128.0 + 45.0 * t + rng.normal(0, 2.0, t.shape)), 0, 255).astype(np.uint8)
        img[h - bar:, :] = bar_px
        tiles.append(img); offs.append((y0, x0))
    truth = [[offs[k + 1][0] - offs[k][0], offs[k + 1][1] - offs[k][1]] for k in range(n - 1)]
    return tiles, truth
