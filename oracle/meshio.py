"""CPU oracle of the preprocess I/O path (TEST INFRASTRUCTURE ONLY; never imported by the product).

* read_obj: the reference's own parsing loop (gen_utils.py:211-226) restated line for line in Python -- pinned against
  the reference function itself by tests/golden/make_golden_r2_io.py.
* vertex_normals: numpy restatement of open3d's TriangleMesh::ComputeVertexNormals, the call gen_utils.py:231 makes.
  open3d is not installed in the build container, so this restatement is NOT pinned against it: PARITY UNPINNED.
* preprocess_scan: preprocess_data.py:38-58 restated in numpy (pinned against the reference script run on a synthetic
  scan, same golden script).
"""
import numpy as np

Y_AXIS_MAX = 33.15232091532151
Y_AXIS_MIN = -36.9843781139949


def read_obj(path):
    vertex_ls, tri_ls = [], []
    with open(path, "r") as f:
        while True:
            line = f.readline().split()
            if not line:
                break
            if line[0] == "v":
                vertex_ls.append(list(map(float, line[1:4])))
            elif line[0] == "f":
                t = list(map(str, line[1:4]))
                if "//" in t[0]:
                    t = [x.split("//")[0] for x in t]
                tri_ls.append(list(map(int, t)))
    return np.array(vertex_ls, dtype=np.float64).reshape(-1, 3), np.array(tri_ls, dtype=np.int64).reshape(-1, 3)


def vertex_normals(vertices, triangles):
    v = np.asarray(vertices, dtype=np.float64)
    t = np.asarray(triangles, dtype=np.int64)
    out = np.zeros_like(v)
    for a, b, c in t:                      # sequential, in file order: the accumulation order open3d uses
        u, w = v[b] - v[a], v[c] - v[a]
        n = np.array([u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]])
        out[a] += n
        out[b] += n
        out[c] += n
    # Eigen's normalize() (>= 3.3) divides only when the squared norm is > 0: a zero sum stays (0, 0, 0)
    z = out[:, 0] * out[:, 0] + out[:, 1] * out[:, 1] + out[:, 2] * out[:, 2]
    pos = z > 0
    out[pos] = out[pos] / np.sqrt(z[pos])[:, None]
    bad = np.isnan(out[:, 0])
    out[bad] = (0.0, 0.0, 1.0)
    return out


def remap_fdi_labels(labels, jaw):
    labels = np.array(labels).reshape(-1, 1)
    if jaw == "lower":
        labels -= 20
    labels[labels // 10 == 1] %= 10
    labels[labels // 10 == 2] = (labels[labels // 10 == 2] % 10) + 8
    labels[labels < 0] = 0
    return labels


def preprocess_scan(vertices_normals, labels, jaw, fps):
    """vertices_normals (n,6) float64, raw FDI labels, fps(xyz (n,3), m) -> indices: the (<=24000, 7) array of :58."""
    vertices = np.array(vertices_normals, dtype=np.float64)
    vertices[:, :3] -= np.mean(vertices[:, :3], axis=0)
    vertices[:, :3] = ((vertices[:, :3] - Y_AXIS_MIN) / (Y_AXIS_MAX - Y_AXIS_MIN)) * 2 - 1
    lv = np.concatenate([vertices, remap_fdi_labels(labels, jaw)], axis=1)
    if lv.shape[0] > 24000:
        lv = lv[np.asarray(fps(lv[:, :3], 24000))[:24000]]
    return lv
