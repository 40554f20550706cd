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

#include <charconv>
#include <mutex>
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
    // Decimal tokens (every coordinate of a real scan): std::from_chars is correctly rounded like float() / strtod and does not
    // go through the locale.  The other spellings float() takes ("inf", "nan", ...) go through strtod below.
    const char *b = t.p, *e = t.p + t.n;
    if (*b == '+') {
        ++b;
        if (b < e && (*b == '-' || *b == '+')) return false;
    }
    char c0 = b < e ? *b : 0;
    if (c0 == '-' && b + 1 < e) c0 = b[1];
    if ((c0 >= '0' && c0 <= '9') || c0 == '.') {
        double r;
        const auto res = std::from_chars(b, e, r);
        if (res.ec == std::errc() && res.ptr == e) {
            *v = r;
            return true;
        }
        if (!(res.ec == std::errc::result_out_of_range && res.ptr == e)) return false;   // "0x10", "1e", "1.2.3", ...
        // overflow / underflow: strtod's inf / 0 below
    }
    for (const char *c = b; c < e; ++c)
        if (*c == 'x' || *c == 'X') return false;                                       // float() takes no hex literals
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
    {   // [+-]digits, at most 18 of them: what every face of a real scan is
        const char *c = t.p, *e = t.p + t.n;
        const bool neg = *c == '-';
        if (*c == '-' || *c == '+') ++c;
        if (c < e && e - c <= 18) {
            long long r = 0;
            for (; c < e && *c >= '0' && *c <= '9'; ++c) r = r * 10 + (*c - '0');
            if (c == e) {
                *v = neg ? -r : r;
                return true;
            }
            return false;   // a non-digit: int() raises
        }
    }
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
                     long long *nv_out, long long *nf_out, std::vector<double> *vsink = nullptr,
                     std::vector<long long> *fsink = nullptr) {
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
            if (vsink) vsink->insert(vsink->end(), c, c + 3);
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
            if (fsink) fsink->insert(fsink->end(), idx, idx + 3);
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


// ---- the ground-truth json of a scan (preprocess_data.py:37-38): {"jaw": "upper" | "lower", "labels": [FDI numbers], ...}.
// A strict reader for exactly that: a top-level object whose "labels" member is an array of plain integers and whose "jaw"
// member is a string without escapes.  Anything else is TGN_ERR_UNSUPPORTED and the caller goes through Python's json.
struct JsonCur {
    const char *s, *e;
    void ws() {
        while (s < e && (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r')) ++s;
    }
    bool eat(char c) {
        ws();
        if (s < e && *s == c) {
            ++s;
            return true;
        }
        return false;
    }
    // s at the opening quote; leaves s behind the closing one.  plain: no backslash inside.
    bool string(const char **b, size_t *n, bool *plain) {
        ws();
        if (s >= e || *s != '"') return false;
        ++s;
        *b = s;
        *plain = true;
        while (s < e && *s != '"') {
            if (*s == '\\') {
                *plain = false;
                ++s;
            }
            ++s;
        }
        if (s >= e) return false;
        *n = (size_t)(s - *b);
        ++s;
        return true;
    }
    // skips any value (nesting by bracket counting, strings honoured)
    bool skip_value() {
        ws();
        if (s >= e) return false;
        if (*s == '"') {
            const char *b;
            size_t n;
            bool plain;
            return string(&b, &n, &plain);
        }
        if (*s == '{' || *s == '[') {
            long depth = 0;
            while (s < e) {
                if (*s == '"') {
                    const char *b;
                    size_t n;
                    bool plain;
                    if (!string(&b, &n, &plain)) return false;
                    continue;
                }
                if (*s == '{' || *s == '[') ++depth;
                if (*s == '}' || *s == ']') {
                    --depth;
                    if (depth == 0) {
                        ++s;
                        return true;
                    }
                }
                ++s;
            }
            return false;
        }
        const char *b = s;
        while (s < e && *s != ',' && *s != '}' && *s != ']' && *s != ' ' && *s != '\t' && *s != '\n' && *s != '\r') ++s;
        return scalar_ok(b, (size_t)(s - b));
    }
    // true / false / null / a JSON number (plus the NaN / Infinity spellings Python's json accepts)
    static bool scalar_ok(const char *b, size_t n) {
        auto is = [&](const char *w) { return strlen(w) == n && !memcmp(b, w, n); };
        if (is("true") || is("false") || is("null") || is("NaN") || is("Infinity") || is("-Infinity")) return true;
        size_t i = 0;
        auto digits = [&]() {
            const size_t i0 = i;
            while (i < n && b[i] >= '0' && b[i] <= '9') ++i;
            return i - i0;
        };
        if (i < n && b[i] == '-') ++i;
        const size_t int0 = i, nd = digits();
        if (nd == 0 || (nd > 1 && b[int0] == '0')) return false;
        if (i < n && b[i] == '.') {
            ++i;
            if (digits() == 0) return false;
        }
        if (i < n && (b[i] == 'e' || b[i] == 'E')) {
            ++i;
            if (i < n && (b[i] == '+' || b[i] == '-')) ++i;
            if (digits() == 0) return false;
        }
        return i == n;
    }
    bool int_array(std::vector<long long> &out) {
        out.clear();
        if (!eat('[')) return false;
        if (eat(']')) return true;
        for (;;) {
            ws();
            bool neg = false;
            if (s < e && *s == '-') {
                neg = true;
                ++s;
            }
            const char *d = s;
            long long v = 0;
            while (s < e && *s >= '0' && *s <= '9') {
                if (s - d >= 18) return false;
                v = v * 10 + (*s - '0');
                ++s;
            }
            if (s == d || (s - d > 1 && *d == '0')) return false;           // no digits / leading zero: not JSON
            if (s < e && (*s == '.' || *s == 'e' || *s == 'E')) return false;  // a float: Python's path decides
            out.push_back(neg ? -v : v);
            if (eat(',')) continue;
            return eat(']');
        }
    }
};

static int parse_scan_json(const std::string &data, std::vector<long long> &labels, std::string &jaw) {
    JsonCur c{data.data(), data.data() + data.size()};
    bool have_labels = false, have_jaw = false;
    if (!c.eat('{')) return TGN_ERR_UNSUPPORTED;
    if (!c.eat('}')) {
        for (;;) {
            const char *kb;
            size_t kn;
            bool plain;
            if (!c.string(&kb, &kn, &plain) || !c.eat(':')) return TGN_ERR_UNSUPPORTED;
            if (!plain) return TGN_ERR_UNSUPPORTED;          // an escaped key could spell "labels" / "jaw"
            if (kn == 6 && !memcmp(kb, "labels", 6)) {
                if (!c.int_array(labels)) return TGN_ERR_UNSUPPORTED;
                have_labels = true;
            } else if (kn == 3 && !memcmp(kb, "jaw", 3)) {
                const char *vb;
                size_t vn;
                if (!c.string(&vb, &vn, &plain) || !plain) return TGN_ERR_UNSUPPORTED;
                for (size_t i = 0; i < vn; ++i)
                    if ((unsigned char)vb[i] < 0x20 || (unsigned char)vb[i] >= 0x7f) return TGN_ERR_UNSUPPORTED;
                jaw.assign(vb, vn);
                have_jaw = true;
            } else if (!c.skip_value()) {
                return TGN_ERR_UNSUPPORTED;
            }
            if (c.eat(',')) continue;
            if (c.eat('}')) break;
            return TGN_ERR_UNSUPPORTED;
        }
    }
    c.ws();
    if (c.s != c.e || !have_labels || !have_jaw) return TGN_ERR_UNSUPPORTED;
    return TGN_OK;
}

static long long floordiv10(long long a) { return a / 10 - ((a % 10 != 0 && a < 0) ? 1 : 0); }

// preprocess_data.py:39-44, in the order numpy applies the four masked assignments
static long long remap_fdi(long long l, bool lower) {
    if (lower) l -= 20;
    if (floordiv10(l) == 1) l %= 10;
    if (floordiv10(l) == 2) l = l % 10 + 8;
    if (l < 0) l = 0;
    return l;
}

// Everything a scan needs on the host, result and scratch alike.  The objects are POOLED: a raw scan touches ~25 MB of fresh heap
// (file text, vertices, faces, normals, rows) -- 4 500 page faults, a third of the loader's time in the kernel -- and with eight
// ranks x several loader threads on one node those faults contend (measured: 20 -> 120 ms per scan and thread from 16 to 64 loaders,
// profiles/r04_preprocess_host_scaling.txt).  A recycled object keeps its vectors' capacity: no allocation, no fault after warm-up.
struct Scan {
    std::vector<double> labeled;   // (n, 7)
    std::string jaw;
    long long n = 0;
    std::string data;              // scratch: file text
    std::vector<long long> labels, f;
    std::vector<double> v, nrm;
};
constexpr size_t kScanPoolMax = 64;   // objects kept (each retains ~25 MB of capacity for a 100 000-vertex scan)
static std::mutex g_scan_mu;
static std::vector<Scan *> g_scan_pool;

static Scan *scan_acquire() {
    {
        std::lock_guard<std::mutex> lock(g_scan_mu);
        if (!g_scan_pool.empty()) {
            Scan *s = g_scan_pool.back();
            g_scan_pool.pop_back();
            return s;
        }
    }
    return new Scan();
}
static void scan_release(Scan *s) {
    {
        std::lock_guard<std::mutex> lock(g_scan_mu);
        if (g_scan_pool.size() < kScanPoolMax) {
            g_scan_pool.push_back(s);
            return;
        }
    }
    delete s;
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

// One scan of the preprocess loop, preprocess_data.py:37-52 in one GIL-free call: the ground-truth json (jaw, FDI labels ->
// 0..16), the OBJ (vertices + open3d-style normals), the centring and the fixed-range scaling, concatenated to the (n, 7)
// float64 rows [x y z nx ny nz label] the reference holds in `labeled_vertices` before sampling.  The arithmetic repeats
// numpy's: np.mean(axis=0) of a C-ordered (n, 3) view is a row-by-row running sum per column, and
// ((v - mean) - Ymin) / (Ymax - Ymin) * 2 - 1 is evaluated per element in that order.
// Returns TGN_ERR_UNSUPPORTED for a json this strict reader does not take (the caller then uses Python's json module) and
// TGN_ERR_INVALID_ARGUMENT where the reference raises (unreadable file, bad OBJ token, label count != vertex count).
TGN_API int tgn_scan_open(const char *obj_path, const char *json_path, double y_min, double y_max, void **handle,
                          long long *n_vertices, char *jaw, int jaw_cap) {
    if (!obj_path || !json_path || !handle || !n_vertices || !jaw || jaw_cap < 2) {
        set_error("tgn_scan_open: bad argument");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    *handle = nullptr;
    Scan *sc = scan_acquire();
    struct Guard {   // back to the pool on every early return
        Scan *s;
        ~Guard() {
            if (s) scan_release(s);
        }
    } guard{sc};
    std::string &data = sc->data;
    std::vector<long long> &labels = sc->labels, &f = sc->f;
    std::vector<double> &v = sc->v, &nrm = sc->nrm;
    labels.clear();
    v.clear();
    f.clear();
    if (int rc = read_file(json_path, data)) return rc;
    std::string jaw_s;
    if (parse_scan_json(data, labels, jaw_s) != TGN_OK || (long long)jaw_s.size() >= jaw_cap) {
        set_error("tgn_scan_open: %s is not a plain {\"jaw\": str, \"labels\": [int]} object", json_path);
        return TGN_ERR_UNSUPPORTED;
    }
    if (int rc = read_file(obj_path, data)) return rc;
    long long nv = 0, nf = 0;
    v.reserve(data.size() / 24);                                // ~ "v -12.345678 -12.345678 -12.345678\n" per vertex, two
    f.reserve(data.size() / 12);                                //   "f 123456 123457 123458\n" per vertex: a first guess
    if (int rc = parse_obj(data, nullptr, nullptr, 0, 0, &nv, &nf, &v, &f)) return rc;
    nrm.resize((size_t)nv * 3);
    for (auto &x : f) x -= 1;                                   // gen_utils.py:226
    if (int rc = tgn_vertex_normals(v.data(), nv, f.data(), nf, nrm.data())) return rc;
    if ((long long)labels.size() != nv) {
        set_error("tgn_scan_open: %lld labels for %lld vertices (np.concatenate raises)", (long long)labels.size(), nv);
        return TGN_ERR_INVALID_ARGUMENT;
    }
    sc->n = nv;
    sc->jaw = jaw_s;
    sc->labeled.resize((size_t)nv * 7);
    double sum[3] = {0.0, 0.0, 0.0};
    for (long long i = 0; i < nv; ++i)
        for (int a = 0; a < 3; ++a) sum[a] += v[i * 3 + a];
    const double mean[3] = {sum[0] / (double)nv, sum[1] / (double)nv, sum[2] / (double)nv};
    const double range = y_max - y_min;
    const bool lower = jaw_s == "lower";
    for (long long i = 0; i < nv; ++i) {
        double *row = sc->labeled.data() + i * 7;
        for (int a = 0; a < 3; ++a) {
            const double centred = v[i * 3 + a] - mean[a];
            const double unit = (centred - y_min) / range;
            row[a] = unit * 2.0 - 1.0;                          // (x * 2 is exact, so a contracted fma gives the same bits)
            row[3 + a] = nrm[i * 3 + a];
        }
        row[6] = (double)remap_fdi(labels[i], lower);
    }
    memcpy(jaw, jaw_s.c_str(), jaw_s.size() + 1);
    *n_vertices = nv;
    *handle = sc;
    guard.s = nullptr;   // the caller owns it until tgn_scan_take
    return TGN_OK;
}

// Frees the pooled scan objects (their retained scratch: ~25 MB each); tgn_scan_open refills the pool on demand.  Returns how many.
TGN_API int tgn_scan_pool_trim(void) {
    std::vector<Scan *> drop;
    {
        std::lock_guard<std::mutex> lock(g_scan_mu);
        drop.swap(g_scan_pool);
    }
    for (Scan *sc : drop) delete sc;
    return (int)drop.size();
}

// Copies the (n, 7) rows out (and, when xyz32 is given, the float32 copy of the coordinates the sampler takes) and frees
// the handle (back to the pool).  With both pointers null it only frees.
TGN_API int tgn_scan_take(void *handle, double *labeled, float *xyz32) {
    Scan *sc = (Scan *)handle;
    if (!sc) {
        set_error("tgn_scan_take: null handle");
        return TGN_ERR_INVALID_ARGUMENT;
    }
    if (labeled) memcpy(labeled, sc->labeled.data(), sc->labeled.size() * sizeof(double));
    if (xyz32)
        for (long long i = 0; i < sc->n; ++i)
            for (int a = 0; a < 3; ++a) xyz32[i * 3 + a] = (float)sc->labeled[i * 7 + a];
    scan_release(sc);
    return TGN_OK;
}
