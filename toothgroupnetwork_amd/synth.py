"""Synthetic scans for tests and bench.py (there is no network and no 3DTeethSeg data here).

Shapes follow SURVEY.md section 8(d): the preprocessed scans of the reference are
(24000, 6) fp32 = xyz in about [-1, 1] (preprocess_data.py:48-50) + unit normals
(generator.py:44).

uniform : xyz ~ U[-1,1]^3 -- stresses FPS, leaves r=0.05 balls nearly empty.
arch    : points on a 2-D horseshoe ("dental arch") surface with per-tooth bumps, about
          6000 pts per unit area at 24000 points so r=0.05 balls hold ~45 points: the
          realistic ball-query / grouping load.
lattice : integer lattice scaled into [-1,1] plus duplicated vertices -- exact distance ties.
"""
import numpy as np


def uniform_cloud(n, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3), dtype=np.float32) * 2.0 - 1.0).astype(np.float32)


def _arch_surface(t, s):
    """t in [0,1] along the arch, s in [0,1] across the crown profile -> (x, y, z)."""
    ang = np.pi * t
    cx, cy = 0.72 * np.cos(ang), 0.85 * np.sin(ang) - 0.45
    # outward normal of the centre line in the xy plane
    nx, ny = 0.85 * np.cos(ang), 0.72 * np.sin(ang)
    nl = np.sqrt(nx * nx + ny * ny)
    nx, ny = nx / nl, ny / nl
    phi = np.pi * s
    bump = 1.0 + 0.18 * np.sin(14.0 * np.pi * t) ** 2
    w = 0.24 * np.cos(phi)
    h = 0.42 * np.sin(phi) * bump
    return cx + nx * w, cy + ny * w, h - 0.2


def arch_cloud(n, seed=0, with_normals=True):
    """(n, 6) fp32: xyz + unit normal (or (n,3) if with_normals=False), in random vertex order."""
    rng = np.random.default_rng(seed)
    t = rng.random(n)
    s = rng.random(n)
    x, y, z = _arch_surface(t, s)
    xyz = np.stack([x, y, z], axis=1)
    xyz += rng.normal(scale=0.0015, size=xyz.shape)
    if not with_normals:
        return xyz.astype(np.float32)
    e = 1e-4
    pt = np.stack(_arch_surface(t + e, s), axis=1) - np.stack(_arch_surface(t - e, s), axis=1)
    ps = np.stack(_arch_surface(t, s + e), axis=1) - np.stack(_arch_surface(t, s - e), axis=1)
    nrm = np.cross(pt, ps)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    return np.concatenate([xyz, nrm], axis=1).astype(np.float32)


def lattice_cloud(side, dup=0, seed=0, shuffle=True):
    """side^3 lattice points in [-1,1]^3 (+ `dup` duplicated vertices): many exact distance ties."""
    g = np.linspace(-1.0, 1.0, side, dtype=np.float32)
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    rng = np.random.default_rng(seed)
    if dup:
        xyz = np.concatenate([xyz, xyz[rng.integers(0, xyz.shape[0], size=dup)]], axis=0)
    if shuffle:
        xyz = xyz[rng.permutation(xyz.shape[0])]
    return np.ascontiguousarray(xyz, dtype=np.float32)


def scan_batch(b, n, kind="arch", seed=0):
    """(b, n, 6) fp32 feature tensor (xyz + normals) as the reference's generator yields per scan."""
    out = np.empty((b, n, 6), dtype=np.float32)
    for i in range(b):
        if kind == "arch":
            out[i] = arch_cloud(n, seed=seed + i)
        else:
            xyz = uniform_cloud(n, seed=seed + i)
            rng = np.random.default_rng(10_000 + seed + i)
            nrm = rng.normal(size=(n, 3)).astype(np.float32)
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            out[i] = np.concatenate([xyz, nrm], axis=1)
    return out


def obj_text(n_u, n_v, seed=0, style="plain", with_tail=True):
    """Text of a synthetic arch-surface OBJ mesh: n_u * n_v vertices on a bumpy horseshoe in millimetre-like units
    (what raw 3DTeethSeg scans look like before preprocess_data.py:48-50 scales them), 2*(n_u-1)*(n_v-1) triangles.
    style "plain": `f a b c`; "slashes": `f a//a b//b c//c` with `vn` lines in between (gen_utils.py:221-223).
    A comment, a `g` line and `vt`-like lines are interleaved (all skipped by the reference loop); with_tail appends a
    BLANK line followed by more vertices, which the reference never reads (`if not line: break`, gen_utils.py:216)."""
    rng = np.random.default_rng(seed)
    u = np.linspace(0.0, np.pi, n_u)
    v = np.linspace(-1.0, 1.0, n_v)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    r = 22.0 + 3.0 * vv
    x = r * np.cos(uu) + rng.normal(0, 0.02, uu.shape)
    y = r * np.sin(uu) * 1.3 + rng.normal(0, 0.02, uu.shape)
    z = 4.0 * np.sin(6 * uu) * np.cos(3 * vv) + 2.0 * vv + rng.normal(0, 0.02, uu.shape)
    lines = ["# synthetic arch mesh", "g scan"]
    for i in range(n_u * n_v):
        lines.append("v %.6f %.6f %.6f" % (x.flat[i], y.flat[i], z.flat[i]))
        if style == "slashes":
            lines.append("vn 0.0 0.0 1.0")
        if i % 97 == 0:
            lines.append("vt 0.5 0.5")
    for i in range(n_u - 1):
        for j in range(n_v - 1):
            a, b, c, d = i * n_v + j + 1, i * n_v + j + 2, (i + 1) * n_v + j + 1, (i + 1) * n_v + j + 2
            for tri in ((a, b, c), (b, d, c)):
                if style == "slashes":
                    lines.append("f " + " ".join("%d//%d" % (t, t) for t in tri))
                else:
                    lines.append("f %d %d %d" % tri)
    if with_tail:
        lines += ["", "v 1000.0 1000.0 1000.0", "f 1 2 3"]
    return "\n".join(lines) + "\n"


def fdi_labels(n, jaw, seed=0):
    """n FDI labels of one jaw (0 = gingiva, 11-18 / 21-28 upper, 31-38 / 41-48 lower) for synthetic ground truth."""
    rng = np.random.default_rng(seed)
    base = (10, 20) if jaw == "upper" else (30, 40)
    teeth = np.array([0] + [b + t for b in base for t in range(1, 9)])
    return teeth[rng.integers(0, teeth.size, n)].tolist()
