"""Host-side geometry helpers mirroring what a Hikari scene script uses from Raycore / GeometryBasics:
Transformation constructors (translate, look_at, perspective — pbrt conventions, camera/perspective.jl:56-91),
`normal_mesh(Rect3f(...))`, `Tesselation(Sphere, n)` (test/volpath_integration.jl:40-66).  float32 throughout."""
import numpy as np

f32 = np.float32


def mat4(rows):
    return np.array(rows, dtype=f32).reshape(4, 4)


def identity():
    return np.eye(4, dtype=f32)


def translate(v):
    m = identity()
    m[:3, 3] = np.asarray(v, dtype=f32)
    return m


def scale(x, y, z):
    m = identity()
    m[0, 0], m[1, 1], m[2, 2] = f32(x), f32(y), f32(z)
    return m


def inv(m):
    return np.linalg.inv(m.astype(np.float64)).astype(f32)


def normalize(v):
    v = np.asarray(v, dtype=f32)
    return (f32(1) / np.sqrt(np.dot(v, v), dtype=f32) * v).astype(f32)


def look_at(pos, look, up):
    """pbrt LookAt: returns the world->camera matrix (PerspectiveCamera inverts it, perspective.jl:56-57)."""
    pos = np.asarray(pos, dtype=f32)
    d = normalize(np.asarray(look, dtype=f32) - pos)
    right = normalize(np.cross(normalize(up), d).astype(f32))
    new_up = np.cross(d, right).astype(f32)
    c2w = identity()
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, new_up, d, pos
    return inv(c2w)


def perspective(fov_deg, n, f):
    p = mat4([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]])
    inv_tan = f32(1) / np.tan(np.deg2rad(f32(fov_deg)) / f32(2), dtype=f32)
    return (scale(inv_tan, inv_tan, 1) @ p).astype(f32)


class Mesh:
    """Triangle soup with optional per-vertex normals / uvs: positions[T,3,3], normals[T,3,3]|None, uvs[T,3,2]|None."""

    def __init__(self, positions, normals=None, uvs=None):
        self.positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=f32).reshape(-1, 3, 3)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=f32).reshape(-1, 3, 2)

    @property
    def n_faces(self):
        return self.positions.shape[0]

    def transformed(self, m):
        m = np.asarray(m, dtype=f32)
        p = self.positions.reshape(-1, 3) @ m[:3, :3].T + m[:3, 3]
        n = None
        if self.normals is not None:
            nm = np.linalg.inv(m[:3, :3].astype(np.float64)).T.astype(f32)
            n = self.normals.reshape(-1, 3) @ nm.T
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
        return Mesh(p.astype(f32), n, self.uvs)


def rect3f(origin, widths):
    """normal_mesh(Rect3f(origin, widths)): 6 quads, 12 triangles, flat per-face normals, outward facing."""
    o = np.asarray(origin, dtype=f32)
    w = np.asarray(widths, dtype=f32)
    lo, hi = o, (o + w).astype(f32)
    X0, Y0, Z0 = lo
    X1, Y1, Z1 = hi
    faces = [  # (corner quad counter-clockwise seen from outside, normal)
        ([(X0, Y0, Z0), (X0, Y0, Z1), (X0, Y1, Z1), (X0, Y1, Z0)], (-1, 0, 0)),
        ([(X1, Y0, Z0), (X1, Y1, Z0), (X1, Y1, Z1), (X1, Y0, Z1)], (1, 0, 0)),
        ([(X0, Y0, Z0), (X1, Y0, Z0), (X1, Y0, Z1), (X0, Y0, Z1)], (0, -1, 0)),
        ([(X0, Y1, Z0), (X0, Y1, Z1), (X1, Y1, Z1), (X1, Y1, Z0)], (0, 1, 0)),
        ([(X0, Y0, Z0), (X0, Y1, Z0), (X1, Y1, Z0), (X1, Y0, Z0)], (0, 0, -1)),
        ([(X0, Y0, Z1), (X1, Y0, Z1), (X1, Y1, Z1), (X0, Y1, Z1)], (0, 0, 1)),
    ]
    P, N, U = [], [], []
    quv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    for q, n in faces:
        for tri in ((0, 1, 2), (0, 2, 3)):
            P.append([q[i] for i in tri])
            N.append([n] * 3)
            U.append([quv[i] for i in tri])
    return Mesh(P, N, U)


def quad(p0, p1, p2, p3, normal=None):
    """Two triangles (p0,p1,p2),(p0,p2,p3)."""
    P = [[p0, p1, p2], [p0, p2, p3]]
    N = None if normal is None else [[normal] * 3, [normal] * 3]
    U = [[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]]
    return Mesh(P, N, U)


def sphere(center, radius, nvertices=32):
    """Tesselation(Sphere(center, r), n): n x n (theta, phi) grid, smooth normals, uv = (phi/2pi, theta/pi)."""
    n = int(nvertices)
    c = np.asarray(center, dtype=f32)
    theta = np.linspace(0, np.pi, n, dtype=np.float64)
    phi = np.linspace(0, 2 * np.pi, n, dtype=np.float64)
    T, Ph = np.meshgrid(theta, phi, indexing="ij")
    nrm = np.stack([np.sin(T) * np.cos(Ph), np.sin(T) * np.sin(Ph), np.cos(T)], axis=-1).astype(f32)
    pos = (c + f32(radius) * nrm).astype(f32)
    uv = np.stack([Ph / (2 * np.pi), T / np.pi], axis=-1).astype(f32)
    P, N, U = [], [], []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, cc, d = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)
            for tri in ((a, b, cc), (a, cc, d)):
                tp = [pos[t] for t in tri]
                e1, e2 = tp[1] - tp[0], tp[2] - tp[0]
                if np.linalg.norm(np.cross(e1, e2)) < 1e-12:
                    continue  # degenerate pole triangle
                P.append(tp)
                N.append([nrm[t] for t in tri])
                U.append([uv[t] for t in tri])
    return Mesh(P, N, U)


def cylinder(p0, p1, radius, nseg=16):
    p0 = np.asarray(p0, dtype=np.float64)
    p1 = np.asarray(p1, dtype=np.float64)
    ax = p1 - p0
    ax /= np.linalg.norm(ax)
    t = np.cross(ax, [1, 0, 0] if abs(ax[0]) < 0.9 else [0, 1, 0])
    t /= np.linalg.norm(t)
    b = np.cross(ax, t)
    P, N = [], []
    for k in range(nseg):
        a0, a1 = 2 * np.pi * k / nseg, 2 * np.pi * (k + 1) / nseg
        n0 = np.cos(a0) * t + np.sin(a0) * b
        n1 = np.cos(a1) * t + np.sin(a1) * b
        q = [p0 + radius * n0, p0 + radius * n1, p1 + radius * n1, p1 + radius * n0]
        qn = [n0, n1, n1, n0]
        for tri in ((0, 1, 2), (0, 2, 3)):
            P.append([q[i] for i in tri])
            N.append([qn[i] for i in tri])
    return Mesh(np.array(P, dtype=f32), np.array(N, dtype=f32), None)
