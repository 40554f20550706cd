// meshio.hip -- host-side (CPU) mesh input of the preprocess path, SURVEY.md 8(f)4: the text-OBJ reader of
// gen_utils.read_txt_obj_ls (gen_utils.py:207-226; "#TODO slow processing speed" in the inference pipelines) and the
// vertex normals its open3d call produces (gen_utils.py:228-233).  No device code: it lives in libtgn_pointops.so so
// that the whole preprocess path binds one C ABI.
//
// Reader semantics are the reference loop's, quirks included:
//   * a line is split on ASCII whitespace; only lines whose FIRST token is exactly "v" or "f" count ("vn", "vt", "#",
//     "g", ... are skipped);
//   * reading STOPS at the first line without any token (an empty or blank line) -- `if not line: break` (:216);
//   * "v": tokens 1..3 parsed as Python float() would (strtod; full token must be consumed);
//   * "f": tokens 1..3; if the first contains "//", every token is cut at its first "//" (:221-223); the result must be
//     a plain integer (int() raises on "1/2/3": status TGN_ERR_INVALID_ARGUMENT here); indices stay 1-based as in the file.
// Vertex normals restate open3d's TriangleMesh::ComputeVertexNormals (0.13+: area-weighted sum of cross(v1-v0, v2-v0)
// over the triangles in file order, then normalisation, (0,0,1) where the result is NaN) in double precision.  open3d is
// not installed in the build container: PARITY UNPINNED for the normals.
#include "tgn_common.h"

#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace tgn {

static bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

struct Tok {
    const char *p;
    size_t n;
};

// tokens of the line [s, e); returns their number (at most cap are stored)
static size_t split_ws(const char *s, const char *e, Tok *out, size_t cap) {
    size_t k = 0;
    while (s < e) {
        while (s < e && is_ws(*s)) ++s;
        if (s >= e) break;
        const char *b = s;
        while (s < e && !is_ws(*s)) ++s;
        if (k < cap) out[k] = Tok{b, (size_t)(s - b)};
        ++k;
    }
    return k;
}

static bool parse_double(const Tok &t, double *v) {
    if (t.n == 0 || t.n > 400) return false;
    char buf[408];
    memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    char *end = nullptr;
    errno = 0;
    *v = strtod(buf, &end);
    return end == buf + t.n;
}

static bool parse_face_index(Tok t, bool cut, long long *v) {
    if (cut) {
        for (size_t i = 0; i + 1 < t.n; ++i)
            if (t.p[i] == '/' && t.p[i + 1] == '/') {
                t.n = i;
                break;
            }
    }
    if (t.n == 0 || t.n > 60) return false;
    char buf[64];
    memcpy(buf, t.p, t.n);
    buf[t.n] = 0;
    char *end = nullptr;
    errno = 0;
    *v = strtoll(buf, &end, 10);
    return end == buf + t.n && errno == 0;
}

static bool has_double_slash(const Tok &t) {
    for (size_t i = 0; i + 1 < t.n; ++i)
        if (t.p[i] == '/' && t.p[i + 1] == '/') return true;
    return false;
}

static int read_file(const char *path, std::string &data) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        set_error("tgn_obj_read: cannot open %s: %s", path, strerror(errno));
        return TGN_ERR_INVALID_ARGUMENT;
    }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && fread(&data[0], 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        set_error("tgn_obj_read: short read on %s", path);
        return TGN_ERR_INVALID_ARGUMENT;
    }
    fclose(f);
    return TGN_OK;
}

// One pass over the text; vertices / faces may be null (counting pass).
static int parse_obj(const std::string &data, double *vertices, long long *faces, long long cap_v, long long cap_f,
                     long long *nv_out, long long *nf_out) {
    const char *s = data.data(), *end = s + data.size();
    long long nv = 0, nf = 0, line_no = 0;
    while (s < end) {
        const char *e = (const char *)memchr(s, '\n', (size_t)(end - s));
        const char *next = e ? e + 1 : end;
        if (!e) e = end;
        ++line_no;
        Tok t[4];
        const size_t k = split_ws(s, e, t, 4);
        if (k == 0) break;   // `if not line: break`
        if (t[0].n == 1 && t[0].p[0] == 'v') {
            if (k < 4) {
                set_error("tgn_obj_read: line %lld: a vertex needs three coordinates", line_no);
                return TGN_ERR_INVALID_ARGUMENT;
            }
            double c[3];
            for (int i = 0; i < 3; ++i)
                if (!parse_double(t[1 + i], &c[i])) {
                    set_error("tgn_obj_read: line %lld: could not convert a coordinate to float", line_no);
                    return TGN_ERR_INVALID_ARGUMENT;
                }
            if (vertices) {
                if (nv >= cap_v) {
                    set_error("tgn_obj_read: more vertices than the buffer holds");
                    return TGN_ERR_INVALID_ARGUMENT;
                }
                vertices[nv * 3 + 0] = c[0];
                vertices[nv * 3 + 1] = c[1];
                vertices[nv * 3 + 2] = c[2];
            }
            ++nv;
        } else if (t[0].n == 1 && t[0].p[0] == 'f') {
            if (k < 4) {
                set_error("tgn_obj_read: line %lld: a face needs three vertex references", line_no);
                return TGN_ERR_INVALID_ARGUMENT;
            }
            const bool cut = has_double_slash(t[1]);
            long long idx[3];
            for (int i = 0; i < 3; ++i)
                if (!parse_face_index(t[1 + i], cut, &idx[i])) {
                    set_error("tgn_obj_read: line %lld: invalid literal for int() in a face", line_no);
                    return TGN_ERR_INVALID_ARGUMENT;
                }
            if (faces) {
                if (nf >= cap_f) {
                    set_error("tgn_obj_read: more faces than the buffer holds");
                    return TGN_ERR_INVALID_ARGUMENT;
                }
                faces[nf * 3 + 0] = idx[0];
                faces[nf * 3 + 1] = idx[1];
                faces[nf * 3 + 2] = idx[2];
            }
            ++nf;
        }
        s = next;
    }
    *nv_out = nv;
    *nf_out = nf;
    return TGN_OK;
}

}  // namespace tgn

using namespace tgn;

// Counts the vertices and faces tgn_obj_read would return.
TGN_API int tgn_obj_count(const char *path, long long *n_vertices, long long *n_faces) {
    if (!path || !n_vertices || !n_faces) {
        set_error("tgn_obj_count: null argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    std::string data;
    if (int rc = read_file(path, data)) return rc;
    return parse_obj(data, nullptr, nullptr, 0, 0, n_vertices, n_faces);
}

// vertices: (cap_v, 3) doubles; faces: (cap_f, 3) int64, 1-based as in the file (HOST pointers).
TGN_API int tgn_obj_read(const char *path, double *vertices, long long *faces, long long cap_v, long long cap_f,
                         long long *n_vertices, long long *n_faces) {
    if (!path || !vertices || !faces || !n_vertices || !n_faces) {
        set_error("tgn_obj_read: null argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    std::string data;
    if (int rc = read_file(path, data)) return rc;
    return parse_obj(data, vertices, faces, cap_v, cap_f, n_vertices, n_faces);
}

// vertices (nv,3) doubles, triangles (nf,3) int64 ZERO-based, normals (nv,3) doubles out (HOST pointers).
TGN_API int tgn_vertex_normals(const double *vertices, long long nv, const long long *triangles, long long nf,
                               double *normals) {
    if (nv < 0 || nf < 0 || (nv > 0 && (!vertices || !normals)) || (nf > 0 && !triangles)) {
        set_error("tgn_vertex_normals: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    for (long long i = 0; i < nv * 3; ++i) normals[i] = 0.0;
    for (long long t = 0; t < nf; ++t) {
        const long long a = triangles[t * 3 + 0], b = triangles[t * 3 + 1], c = triangles[t * 3 + 2];
        if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) {
            set_error("tgn_vertex_normals: triangle %lld references a vertex outside [0, %lld)", t, nv);
            return TGN_ERR_INVALID_ARGUMENT;
        }
        const double *pa = vertices + a * 3, *pb = vertices + b * 3, *pc = vertices + c * 3;
        const double ux = pb[0] - pa[0], uy = pb[1] - pa[1], uz = pb[2] - pa[2];
        const double vx = pc[0] - pa[0], vy = pc[1] - pa[1], vz = pc[2] - pa[2];
        const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;   // v01.cross(v02), unnormalised
        for (const long long k : {a, b, c}) {
            normals[k * 3 + 0] += nx;
            normals[k * 3 + 1] += ny;
            normals[k * 3 + 2] += nz;
        }
    }
    for (long long i = 0; i < nv; ++i) {
        double *n = normals + i * 3;
        // open3d: vertex_normals_[i].normalize(); if (isnan(x)) -> (0, 0, 1).  Eigen (>= 3.3) normalises only when the squared
        // norm is > 0, so a vertex without triangles (or a degenerate fan) KEEPS its zero sum; only NaN input reaches the
        // substitution.  (Restated from the published sources: open3d is not installed here -- parity unpinned.)
        const double z = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (z > 0.0) {
            const double len = sqrt(z);
            n[0] /= len;
            n[1] /= len;
            n[2] /= len;
        }
        if (n[0] != n[0]) {
            n[0] = 0.0;
            n[1] = 0.0;
            n[2] = 1.0;
        }
    }
    return TGN_OK;
}
