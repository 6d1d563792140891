"""Deterministic procedural meshes standing in for the Stanford bunny (no .ply ships with the reference and
there is no network): a geodesic sphere displaced by low-order polynomials of the direction, written as the ASCII
PLY dialect Model::load_ply accepts (reference src/primitives.cpp:238-319).

Only + - * / and sqrt are used (all correctly rounded in IEEE double), so the files are byte-identical on every
machine. Faces are wound counter-clockwise seen from outside: the reference never flips a normal towards the ray,
so an inward-wound mesh shadows itself.
"""
import numpy as np

_PHI = (1.0 + np.sqrt(5.0)) / 2.0
_ICO_V = np.array([[-1, _PHI, 0], [1, _PHI, 0], [-1, -_PHI, 0], [1, -_PHI, 0],
                   [0, -1, _PHI], [0, 1, _PHI], [0, -1, -_PHI], [0, 1, -_PHI],
                   [_PHI, 0, -1], [_PHI, 0, 1], [-_PHI, 0, -1], [-_PHI, 0, 1]], dtype=np.float64)
_ICO_F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                   [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                   [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                   [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)


def geodesic_sphere(n):
    """Unit sphere with 20*n*n faces (every icosahedron face cut into n*n). Returns (V,3) float64, (F,3) int64."""
    pts, faces = [], []
    base = 0
    for fa, fb, fc in _ICO_F:
        A, B, Cc = _ICO_V[fa], _ICO_V[fb], _ICO_V[fc]
        idx = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                k = n - i - j
                idx[(i, j)] = base + len(idx)
                pts.append((k * A + i * B + j * Cc) / n)
        for i in range(n):
            for j in range(n - i):
                faces.append((idx[(i, j)], idx[(i + 1, j)], idx[(i, j + 1)]))
                if i + j < n - 1:
                    faces.append((idx[(i + 1, j)], idx[(i + 1, j + 1)], idx[(i, j + 1)]))
        base += len(idx)
    P = np.array(pts, dtype=np.float64)
    F = np.array(faces, dtype=np.int64)
    # merge the vertices shared along icosahedron edges (exact integer key on a fine lattice)
    key = np.round(P * (1 << 20)).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)              # keep first-seen order: deterministic vertex numbering
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    P = P[first[order]]
    F = rank[inv.reshape(-1)][F]
    P = P / np.sqrt((P * P).sum(axis=1, keepdims=True))
    return P, F


def _lcg(seed, n):
    """n floats in [0,1) from a 64-bit LCG (Knuth MMIX constants): byte-reproducible everywhere."""
    out = np.empty(n, np.float64)
    x = (int(seed) * 2654435761 + 12345) & 0xFFFFFFFFFFFFFFFF
    for i in range(n):
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = (x >> 11) / float(1 << 53)
    return out


def blob(n, seed=1, radius=0.075):
    """Closed, star-shaped, smoothly displaced sphere with 20*n*n faces. Returns float32 vertices, int faces."""
    P, F = geodesic_sphere(n)
    c = _lcg(seed, 8) - 0.5
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    r = (1.0 + 0.30 * c[0] * (3 * x * x * y - y * y * y) + 0.30 * c[1] * (3 * z * z * x - x * x * x)
         + 0.25 * c[2] * (x * y * z * 4) + 0.25 * c[3] * (z * z * z * z - 0.4) + 0.20 * c[4] * (x * x - y * y)
         + 0.15 * (y * y * y) + 0.10 * c[5] * (x * z) + 0.10 * c[6] * (y * z))
    V = (P * (radius * r)[:, None]).astype(np.float32)
    return V, F


def bunny_class(faces=69451, seed=2):
    """Stand-in for bun_zipper.ply (69,451 faces): the geodesic blob whose face count is closest."""
    n = max(1, int(round(np.sqrt(faces / 20.0))))
    return blob(n, seed=seed)


def write_ply(path, V, F):
    """ASCII PLY with 9 significant digits (float32 round-trips exactly)."""
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment raytracer_amd procedural mesh\n")
        f.write(f"element vertex {V.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n")
        f.write(f"element face {F.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n")
        for v in V:
            f.write("%.9g %.9g %.9g\n" % (float(v[0]), float(v[1]), float(v[2])))
        for a, b, c in F:
            f.write("3 %d %d %d\n" % (int(a), int(b), int(c)))


def blob_field(n_blobs, n, seed=3, spacing=0.4, radius=0.12):
    """Many blobs on a square grid in the xz-plane (the 10M-triangle field of BASELINE config 5 is
    n_blobs=2000, n=16). Returns float32 vertices and int faces of ONE mesh."""
    side = int(np.ceil(np.sqrt(n_blobs)))
    Vs, Fs = [], []
    off = 0
    jit = _lcg(seed, 2 * n_blobs) - 0.5
    V0, F0 = blob(n, seed=seed, radius=radius)
    for b in range(n_blobs):
        gx, gz = b % side, b // side
        t = np.array([(gx - side / 2 + jit[2 * b] * 0.3) * spacing, 0.0, (gz - side / 2 + jit[2 * b + 1] * 0.3) * spacing])
        Vs.append((V0.astype(np.float64) + t).astype(np.float32))
        Fs.append(F0 + off)
        off += V0.shape[0]
    return np.concatenate(Vs), np.concatenate(Fs)
