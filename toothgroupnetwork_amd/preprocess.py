"""The data either side of preprocess_data.py's FPS and the sharded preprocess runner of BASELINE.json
config 5 ("preprocess_data.py over the full set sharded across 8 x MI355X, RCCL gather only").

Reference path per scan (preprocess_data.py:35-58): load the ground-truth json, remap FDI labels (:38-44), parse the
text OBJ and take open3d's vertex normals (gen_utils.read_txt_obj_ls, gen_utils.py:201-240), centre, scale by the
fixed Y range (:48-50), append the labels, farthest-point-sample to 24 000 points if the scan has more (:55-56) and
np.save the (24000, 7) float64 array (:58).  After the model, the inference pipeline transfers the 24 000 predicted
labels back to every vertex with a 1-nearest-neighbour query (inference_pipeline_sem.py:37-39).

Here: the OBJ reader and the normals are native (libtgn_pointops.so, include/tgn_pointops.h section 4: the reference's
Python loop is its own "#TODO slow processing speed"), the FPS of a whole shard of scans is ONE launch (one workgroup /
cooperative group per scan, resample.fps_batch) and the label transfer is the GPU kNN with k = 1.  ``preprocess_sharded``
splits the scan list over the ranks of a torch.distributed job: nothing is exchanged while working, one all_gather of
a small fp64 vector at the end.
"""
import ctypes
import json
import os
import threading
import time

import numpy as np

from . import _lib, launch, sharding

Y_AXIS_MAX = 33.15232091532151       # preprocess_data.py:16-17
Y_AXIS_MIN = -36.9843781139949
N_SAMPLED = 24000                    # preprocess_data.py:55


def read_obj(path):
    """(vertices (n,3) float64, faces (m,3) int64, 1-based) of a text OBJ, with the reference loop's semantics
    (gen_utils.py:211-226): see tgn_obj_read in include/tgn_pointops.h.  Raises ValueError where float() / int() would."""
    L = _lib.lib()
    nv, nf = ctypes.c_longlong(0), ctypes.c_longlong(0)
    bpath = os.fsencode(path)
    if L.tgn_obj_count(bpath, ctypes.byref(nv), ctypes.byref(nf)):
        raise ValueError(L.tgn_last_error().decode("utf-8", "replace"))
    v = np.empty((max(nv.value, 0), 3), dtype=np.float64)
    f = np.empty((max(nf.value, 0), 3), dtype=np.int64)
    if L.tgn_obj_read(bpath, v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p), v.shape[0], f.shape[0],
                      ctypes.byref(nv), ctypes.byref(nf)):
        raise ValueError(L.tgn_last_error().decode("utf-8", "replace"))
    return v, f


def vertex_normals(vertices, triangles):
    """open3d-style vertex normals (gen_utils.py:228-233): vertices (n,3) float64, triangles (m,3) ZERO-based -> (n,3)."""
    L = _lib.lib()
    v = np.ascontiguousarray(vertices, dtype=np.float64)
    t = np.ascontiguousarray(triangles, dtype=np.int64)
    out = np.empty_like(v)
    if L.tgn_vertex_normals(v.ctypes.data_as(ctypes.c_void_p), v.shape[0], t.ctypes.data_as(ctypes.c_void_p), t.shape[0],
                            out.ctypes.data_as(ctypes.c_void_p)):
        raise ValueError(L.tgn_last_error().decode("utf-8", "replace"))
    return out


def read_txt_obj_ls(path, ret_mesh=False, use_tri_mesh=False):
    """gen_utils.read_txt_obj_ls (gen_utils.py:201-240): [ (n,6) float64 = vertices + vertex normals ]; with ret_mesh also
    a dict {"vertices", "triangles" (zero-based), "vertex_normals"} standing in for the open3d mesh object."""
    if use_tri_mesh:
        raise NotImplementedError("the trimesh loader of the reference (gen_utils.py:203-206) is not part of this path")
    v, f = read_obj(path)
    tri = f - 1
    norms = vertex_normals(v, tri)
    out = [np.concatenate([v, norms], axis=1)]
    if ret_mesh:
        out.append({"vertices": v, "triangles": tri, "vertex_normals": norms})
    return out


def remap_fdi_labels(labels, jaw):
    """FDI tooth numbers -> 0..16 (preprocess_data.py:39-44); labels: int array, reshaped to (n,1)."""
    labels = np.array(labels).reshape(-1, 1)
    if jaw == "lower":
        labels -= 20
    labels[labels // 10 == 1] %= 10
    labels[labels // 10 == 2] = (labels[labels // 10 == 2] % 10) + 8
    labels[labels < 0] = 0
    return labels


def normalise_vertices(vertices):
    """centre, then map the fixed Y range to [-1, 1] on every axis (preprocess_data.py:48-50); in place on a copy."""
    vertices = np.array(vertices, dtype=np.float64)
    vertices[:, :3] -= np.mean(vertices[:, :3], axis=0)
    vertices[:, :3] = ((vertices[:, :3] - Y_AXIS_MIN) / (Y_AXIS_MAX - Y_AXIS_MIN)) * 2 - 1
    return vertices


def sampled_points_name(base_name, jaw):
    return f"{base_name}_{jaw}_sampled_points"       # np.save appends ".npy" (preprocess_data.py:58)


def load_scan(obj_path, json_path):
    """-> (labeled_vertices (n,7) float64 before sampling, base_name, jaw)   [preprocess_data.py:37-52]"""
    base_name = os.path.basename(obj_path).split(".")[0]
    with open(json_path, "r") as st:
        loaded = json.load(st)
    labels = remap_fdi_labels(loaded["labels"], loaded["jaw"])
    vertices = normalise_vertices(read_txt_obj_ls(obj_path)[0])
    return np.concatenate([vertices, labels], axis=1), str(base_name), loaded["jaw"]


class ArrayPool:
    """Recycled row buffers for the loader threads: a raw scan's (n, 7) float64 rows and (n, 3) float32 coordinates are 7.5 MB of
    fresh memory per scan -- page faults that contend once eight ranks x several loader threads share a node.  take(rows) hands out
    the first `rows` rows of a kept buffer of enough capacity (or a new one, capacity rounded up to 16 384 rows); give(view) takes
    it back."""

    def __init__(self, cols, dtype, keep=160):
        self.cols, self.dtype, self.keep = cols, np.dtype(dtype), keep
        self.lock, self.free = threading.Lock(), []

    def take(self, rows):
        with self.lock:
            for i, buf in enumerate(self.free):
                if buf.shape[0] >= rows:
                    return self.free.pop(i)[:rows]
        cap = -(-max(int(rows), 1) // 16384) * 16384
        return np.empty((cap, self.cols), dtype=self.dtype)[:rows]

    def give(self, view):
        base = view.base if isinstance(view.base, np.ndarray) else view
        with self.lock:
            if len(self.free) < self.keep:
                self.free.append(base)


def load_scan_native(obj_path, json_path, with_xyz32=False, pools=None):
    """load_scan in ONE native call that never holds the interpreter lock (tgn_scan_open / tgn_scan_take): what the sharded
    runner's load threads use, since json.load, the label remap and the numpy glue of load_scan serialise on the GIL.
    -> (labeled_vertices, base_name, jaw, xyz32 or None); xyz32 = float32 copy of the coordinates when with_xyz32 and the
    scan has more than N_SAMPLED vertices.  Returns None when the json is not the plain {"jaw": str, "labels": [int]} shape
    (the caller then takes load_scan); raises ValueError where load_scan raises.  pools: optional (ArrayPool(7, float64),
    ArrayPool(3, float32)) the two arrays are taken from (the caller gives them back)."""
    L = _lib.lib()
    handle, nv = ctypes.c_void_p(), ctypes.c_longlong(0)
    jaw = ctypes.create_string_buffer(64)
    rc = L.tgn_scan_open(os.fsencode(obj_path), os.fsencode(json_path), Y_AXIS_MIN, Y_AXIS_MAX, ctypes.byref(handle),
                         ctypes.byref(nv), jaw, len(jaw))
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    if rc:
        raise ValueError(L.tgn_last_error().decode("utf-8", "replace"))
    try:
        want32 = with_xyz32 and nv.value > N_SAMPLED
        if pools is not None:                                   # (rows64 pool, xyz32 pool): recycled buffers, see ArrayPool
            lv = pools[0].take(nv.value)
            x32 = pools[1].take(nv.value) if want32 else None
        else:
            lv = np.empty((nv.value, 7), dtype=np.float64)
            x32 = np.empty((nv.value, 3), dtype=np.float32) if want32 else None
    except BaseException:
        L.tgn_scan_take(handle, None, None)
        raise
    L.tgn_scan_take(handle, lv.ctypes.data_as(ctypes.c_void_p), x32.ctypes.data_as(ctypes.c_void_p) if x32 is not None else None)
    return lv, str(os.path.basename(obj_path).split(".")[0]), jaw.value.decode("ascii"), x32


_sampler_local = threading.local()


def _default_fps_batch(xyz_list, npoint):
    """resample.fps_batch on a HIP stream of the calling thread's own: the sampler threads of preprocess_scans each keep a
    launch in flight, and launches on different streams share the GPU (one workgroup per scan, 256 CUs)."""
    import torch
    from . import resample
    stream = getattr(_sampler_local, "stream", None)
    if stream is None:
        stream = _sampler_local.stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        return resample.fps_batch(xyz_list, npoint)


def preprocess_scans(pairs, save_path, batch=16, fps_batch=None, workers=None, samplers=None):
    """pairs: [(obj_path, json_path)] -> one "<id>_<jaw>_sampled_points.npy" per scan under save_path, exactly the arrays
    preprocess_data.py writes.  Scans with more than 24 000 vertices are farthest-point-sampled up to `batch` at a time in
    one launch.  Three stages overlap: load threads (`workers`, default min(16, usable host cores / ranks on the node)) run the
    host side of a scan in one native call that never holds the interpreter lock (load_scan_native: json, OBJ parse,
    normals, scaling); `samplers` threads (default 2, TGN_PREPROCESS_SAMPLERS) each pack a batch, launch the FPS on a
    stream of their own and hand the picks to writer threads, which select and save.  The FPS launch is bound by its
    iteration count, not by the number of scans in it (one workgroup per scan), so the loop takes WHATEVER the loaders have
    finished (in submission order, at least batch/4, at most batch): the batch grows until the GPU keeps up with the host
    cores, nobody waits for a fixed-size batch to fill, and two launches in flight share the chip.
    Returns {"scans", "sampled", "points_in", "checksum", "batches", "seconds_load" (time the loop WAITED for loads),
    "seconds_fps" (summed over the sampler threads)}."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    fps_batch = fps_batch or _default_fps_batch
    os.makedirs(save_path, exist_ok=True)
    if workers is None:
        local = max(int(os.environ.get("LOCAL_WORLD_SIZE", "1")), 1)
        # loader threads: the cores this rank can really use (affinity mask and cgroup CPU quota, shared by the ranks of the node),
        # at most 16 -- one process's loaders scale linearly to 16 threads (84 scans/s each) and not beyond
        workers = int(os.environ.get("TGN_PREPROCESS_WORKERS", "0")) or max(1, min(16, sharding.cpus_for_this_rank(local)))
    if samplers is None:
        samplers = max(int(os.environ.get("TGN_PREPROCESS_SAMPLERS", "2")), 1)
    stats = dict(scans=0, sampled=0, points_in=0, checksum=0.0, batches=0, seconds_load=0.0, seconds_fps=0.0)
    step = max(int(batch), 1)
    at_least = max(step // 4, 1)
    pool = ThreadPoolExecutor(max_workers=workers)
    sampler = ThreadPoolExecutor(max_workers=samplers)
    writer = ThreadPoolExecutor(max_workers=4)

    pools = (ArrayPool(7, np.float64, keep=2 * step + workers), ArrayPool(3, np.float32, keep=2 * step + workers))   # (what can be in flight)

    def load(obj_path, json_path):
        fast = load_scan_native(obj_path, json_path, with_xyz32=True, pools=pools)
        if fast is not None:
            return fast
        lv, name, jaw = load_scan(obj_path, json_path)         # a json the strict native reader hands back
        return lv, name, jaw, (np.ascontiguousarray(lv[:, :3], dtype=np.float32) if lv.shape[0] > N_SAMPLED else None)

    def select_and_save(lv, ix, name, jaw):
        # (writer thread) gen_utils.resample_pcd: pcd[idx[:n]], then np.save (preprocess_data.py:56-58)
        out = lv[np.asarray(ix)[:N_SAMPLED]] if ix is not None else lv
        np.save(os.path.join(save_path, sampled_points_name(name, jaw)), out)
        pools[0].give(lv)

    def sample(loaded):
        # (sampler thread) one FPS launch over the scans of this batch that need it, then the writes
        t0 = time.perf_counter()
        big = [i for i, item in enumerate(loaded) if item[3] is not None]
        idx = fps_batch([loaded[i][3] for i in big], N_SAMPLED) if big else []
        for i in big:                                           # (the sampler copied the coordinates into its staging buffer)
            pools[1].give(loaded[i][3])
        picked = dict(zip(big, idx))
        checksum = sum(float(np.asarray(ix, dtype=np.int64).sum()) for ix in idx)
        dt = time.perf_counter() - t0
        writes = [writer.submit(select_and_save, lv, picked.get(i), name, jaw) for i, (lv, name, jaw, _) in enumerate(loaded)]
        return len(loaded), len(big), sum(int(item[0].shape[0]) for item in loaded), checksum, dt, writes

    todo, inflight, sampling, writes = iter(pairs), deque(), deque(), []

    def top_up():                                               # at most 2 * batch scans loaded or loading at any time
        while len(inflight) < 2 * step:
            nxt = next(todo, None)
            if nxt is None:
                return
            inflight.append(pool.submit(load, *nxt))

    def retire(fut):
        n, nbig, pts, checksum, dt, ws = fut.result()
        stats["scans"] += n
        stats["sampled"] += nbig
        stats["points_in"] += pts
        stats["checksum"] += checksum
        stats["batches"] += 1
        stats["seconds_fps"] += dt
        writes.extend(ws)

    try:
        top_up()
        while inflight:
            t0 = time.perf_counter()
            loaded = []
            while inflight and (len(loaded) < at_least or (len(loaded) < step and inflight[0].done())):
                loaded.append(inflight.popleft().result())
            top_up()                                            # parsed while the GPU samples
            stats["seconds_load"] += time.perf_counter() - t0
            while len(sampling) >= samplers:
                retire(sampling.popleft())
            sampling.append(sampler.submit(sample, loaded))
        while sampling:
            retire(sampling.popleft())
        for w in writes:
            w.result()                                                             # (re-raises a failed write)
    finally:
        for f in inflight:
            f.cancel()
        pool.shutdown()
        sampler.shutdown()
        writer.shutdown()
        _lib.lib().tgn_scan_pool_trim()                         # the native loader's recycled scratch (~25 MB per loader thread)
    return stats


def list_scans(source_obj_data_path, source_json_data_path):
    """The (obj, json) pairs preprocess_data.py:21-33 walks, sorted so that every rank sees the same order."""
    from glob import glob
    objs = []
    for dir_path in sorted(x[0] for x in os.walk(source_obj_data_path))[1:]:
        objs += sorted(glob(os.path.join(dir_path, "*.obj")))
    jmap = {}
    for dir_path in sorted(x[0] for x in os.walk(source_json_data_path))[1:]:
        for jp in glob(os.path.join(dir_path, "*.json")):
            jmap[os.path.basename(jp).split(".")[0]] = jp
    return [(o, jmap[os.path.basename(o).split(".")[0]]) for o in objs]


def preprocess_sharded(pairs, save_path, rank, world, batch=16, fps_batch=None, device=None, mode="round_robin"):
    """BASELINE.json config 5: rank `rank` of `world` preprocesses its shard of `pairs` (round robin: raw scans differ in
    size); ONE collective at the end gathers the per-rank counters.  Returns the job totals (identical on every rank):
    {"scans", "sampled", "points_in", "checksum", "seconds" (max over ranks), "per_rank_scans", "fps_launches", "meshes_per_s"}."""
    mine = [pairs[i] for i in sharding.shard_indices(len(pairs), rank, world, mode)]
    sharding.barrier()
    t0 = time.perf_counter()
    st = preprocess_scans(mine, save_path, batch=batch, fps_batch=fps_batch)
    dt = time.perf_counter() - t0
    launch.stage("gather")
    mat = sharding.gather_metrics([st["scans"], st["sampled"], st["points_in"], st["checksum"], dt, st["seconds_load"],
                                   st["seconds_fps"], st["batches"]], device=device).cpu().numpy()
    tot = mat.sum(0)
    seconds = float(mat[:, 4].max())
    return {"scans": int(round(tot[0])), "sampled": int(round(tot[1])), "points_in": int(round(tot[2])), "checksum": float(tot[3]),
            "seconds": seconds, "seconds_load_max": float(mat[:, 5].max()), "seconds_fps_max": float(mat[:, 6].max()),
            "per_rank_scans": [int(round(v)) for v in mat[:, 0]], "fps_launches": int(round(tot[7])), "meshes_per_s": float(tot[0] / seconds) if seconds > 0 else 0.0}


def transfer_labels(sampled_xyz, sampled_labels, vertices, candidates=4):
    """Labels of the 24 000 sampled points back onto every vertex: the 1-nearest-neighbour query of
    inference_pipeline_sem.py:37-39 (sklearn KDTree on float64 coordinates there).  The GPU kNN proposes the `candidates`
    nearest samples per vertex in fp32 (the kernel's arithmetic) and the nearest of those is picked by float64 distances
    on the original coordinates -- the KDTree's answer wherever the nearest sample is unique in float64 (equidistant
    samples: the lower sample index)."""
    import torch
    from . import pointops
    dev = torch.device("cuda")
    s64 = torch.from_numpy(np.ascontiguousarray(np.asarray(sampled_xyz)[:, :3], dtype=np.float64)).to(dev)
    v64 = torch.from_numpy(np.ascontiguousarray(np.asarray(vertices)[:, :3], dtype=np.float64)).to(dev)
    k = max(1, min(int(candidates), s64.shape[0]))
    o = torch.tensor([s64.shape[0]], dtype=torch.int32, device=dev)
    n_o = torch.tensor([v64.shape[0]], dtype=torch.int32, device=dev)
    idx, _ = pointops.knnquery(k, s64.float().contiguous(), v64.float().contiguous(), o, n_o)     # (V, k) int32
    idx = idx.long()
    if k > 1:
        d = ((s64[idx.reshape(-1)].view(-1, k, 3) - v64[:, None, :]) ** 2).sum(-1)               # float64, (V, k)
        best = d.min(1, keepdim=True).values
        cand = torch.where(d == best, idx, torch.full_like(idx, s64.shape[0]))                    # exact ties: lowest index
        idx = cand.min(1).values
    return np.asarray(sampled_labels).reshape(-1)[idx.reshape(-1).cpu().numpy()]
