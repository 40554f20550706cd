"""The reference's per-scan evaluation loop, sharded over the GPUs of one node.

Reference: ``Trainer.test`` (trainer.py:49-54) -- one ``model.step(batch_idx, batch_item, "test")`` per scan of the
validation loader (batch size 1), each step's ``LossMap.get_loss_dict_for_print("val")`` folded into a ``LossMeter``
(loss_meter.py:2-23), ``get_avg_results()`` at the end.  Scans are independent, so rank r of `world` takes every
world-th scan (round robin over the sorted file list), nothing is exchanged while computing, and ONE
``all_gather_into_tensor`` of a small fp64 vector -- step count, seconds, the meter's sums -- combines the ranks
(sharding.gather_metrics: RCCL over xGMI on GPUs, gloo on CPU).  The averages every rank ends up with are the serial
loop's: sum over all scans / number of scans.

The model side is a *step object*: ``step(batch_idx, batch_item) -> {name_val: float, ..., total_val: float}`` with a
fixed ``keys`` tuple (the metric schema must be the same on every rank, also on a rank whose shard is empty).
``PointNetPPStep`` / ``PointTransformerStep`` restate the "test" branch of models/pointnet_pp_model.py:14-39 and
models/transformer_model.py:13-36 over this package's network mirrors (nets.py); they need the GPU like every operator
of the package.  Tests of the control flow hand in their own step object.
"""
import glob
import os
import time

import numpy as np
import torch

from . import launch, sharding


class LossMeter:
    """loss_meter.py:2-23: running sums per key and a step count; averages on request."""

    def __init__(self):
        self.init()

    def init(self):
        self.step_num = 0
        self.loss_meter_dict = {}

    def aggr(self, loss_map):
        for key, value in loss_map.items():
            self.loss_meter_dict[key] = self.loss_meter_dict.get(key, 0) + value
        self.step_num += 1

    def get_avg_results(self):
        return {key: total / self.step_num for key, total in self.loss_meter_dict.items()}


def list_preprocessed(data_dir):
    """the files DentalModelGenerator reads (generator.py:13), sorted so that every rank sees the same order"""
    return sorted(glob.glob(os.path.join(data_dir, "*_sampled_points.npy")))


def load_item(path):
    """One validation item the way generator.py:40-66 builds it (no augmentation), already batched by the loader's
    collate (runner.py:7-19) at batch size 1: feat (1, 6, N) fp32, gt_seg_label (1, 1, N) int64 with gingiva = -1."""
    arr = np.load(path)
    # (layout changes in numpy: torch's CPU kernels would fan a 144 000-element copy out over every hardware thread of the box --
    #  25 ms per scan on a 128-thread host under a 16-core quota, against 3 ms for the forward itself)
    feat = np.ascontiguousarray(arr[:, :6].T, dtype=np.float32)[None]
    seg = np.ascontiguousarray(arr[:, 6:].T.astype(np.int64) - 1)[None]
    return {"feat": torch.from_numpy(feat), "gt_seg_label": torch.from_numpy(seg), "mesh_path": [path]}


def print_dict(named_losses, post_fix):
    """LossMap.get_loss_dict_for_print (loss_meter.py:50-62): value * weight per key, and their total"""
    out = {f"{name}_{post_fix}": float(value) * weight for name, (value, weight) in named_losses.items()}
    out[f"total_{post_fix}"] = sum(out.values())
    return out


def tooth_class_loss(cls_pred, gt_cls):
    """tgn_loss.py:355-367 without weights / smoothing: cross entropy of (B, 17, N) logits against labels shifted by one
    (gingiva -1 -> class 0)"""
    b = gt_cls.shape[0]
    return torch.nn.functional.cross_entropy(cls_pred.float(), gt_cls.view(b, -1).long() + 1)


class _ClassStep:
    keys = ("tooth_class_loss_1_val", "total_val")

    def __init__(self, module, device, weight=1, output_index=0):
        self.module, self.device, self.weight, self.output_index = module.to(device).eval(), device, weight, output_index

    def __call__(self, batch_idx, batch_item):
        points = batch_item["feat"].to(self.device)
        seg_label = batch_item["gt_seg_label"].to(self.device)
        with torch.no_grad():
            output = self.module([points, seg_label])
        loss = tooth_class_loss(output[self.output_index], seg_label)
        return print_dict({"tooth_class_loss_1": (loss.item(), self.weight)}, "val")


class PointNetPPStep(_ClassStep):
    """models/pointnet_pp_model.py:14-39, phase "test": class logits are output 6 of the network (pointnet_pp.py:63-68)"""

    def __init__(self, module, device):
        super().__init__(module, device, weight=1, output_index=6)


class PointTransformerStep(_ClassStep):
    """models/transformer_model.py:13-36, phase "test": `sem_1` is the first output of the segmentation network"""

    def __init__(self, module, device, weight=1):
        super().__init__(module, device, weight=weight, output_index=0)


def reference_state_dict(state):
    """A checkpoint the reference wrote holds `self.module.state_dict()` of its wrapper module (base_model.py:33-37), whose
    network sits under `first_sem_model.` (pointnet_pp.py:78) or `first_ins_cent_model.` (point_transformer.py:9) next to the
    loss's `criterion.*` entries; the network mirrors here (nets.py) are the bare networks.  Strips that prefix and drops the
    criterion entries; a state_dict of a bare network passes through unchanged."""
    prefixes = ("first_sem_model.", "first_ins_cent_model.")
    if not any(k.startswith(prefixes) for k in state):
        return state
    out = {}
    for k, v in state.items():
        if k.startswith("criterion."):
            continue
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out


def eval_sharded(paths, step, rank, world, device=None, load=load_item):
    """Trainer.test's loop over this rank's share of `paths` and the one gather.  Returns, on every rank,
    {"avg": LossMeter averages over ALL scans, "steps": scans evaluated, "per_rank_steps", "per_rank_seconds",
     "scans_per_s": steps / slowest rank's seconds}.
    A scan that fails on one rank (a load error, a step whose keys differ from its schema, a gather index out of range)
    does not leave the other ranks waiting in the collective: the rank stops its loop, carries an error flag in the
    gathered vector, and EVERY rank raises after the gather, naming the ranks that failed."""
    keys = tuple(step.keys)
    meter = LossMeter()
    mine = sharding.shard_indices(len(paths), rank, world, "round_robin")
    sharding.barrier()
    t0 = time.perf_counter()
    failure = None
    for i in mine:
        try:
            loss_map = step(i, load(paths[i]))
            if tuple(sorted(loss_map)) != tuple(sorted(keys)):
                raise KeyError(f"step returned keys {sorted(loss_map)}, its schema says {sorted(keys)}")
        except Exception as e:  # noqa: BLE001
            failure = (i, e)
            launch.note(scan_error=f"scan {i} ({os.path.basename(str(paths[i]))}): {type(e).__name__}: {str(e)[:200]}")
            break
        meter.aggr(loss_map)
    if device is not None and device.type == "cuda":
        try:
            torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001
            failure = failure or (-1, e)
    seconds = time.perf_counter() - t0
    vec = [float(meter.step_num), seconds, 0.0 if failure is None else 1.0] + [float(meter.loss_meter_dict.get(k, 0.0)) for k in keys]
    launch.stage("gather")
    mat = sharding.gather_metrics(vec, device=device).cpu()           # the one collective of the run
    failed = [r for r in range(mat.shape[0]) if float(mat[r, 2]) != 0.0]
    if failed:
        mine_txt = f"; here: scan {failure[0]}: {type(failure[1]).__name__}: {failure[1]}" if failure is not None else ""
        raise RuntimeError(f"evaluation failed on rank(s) {failed}{mine_txt}") from (failure[1] if failure is not None else None)
    steps = int(round(float(mat[:, 0].sum())))
    sums = mat[:, 3:].sum(0).tolist()
    slowest = float(mat[:, 1].max())
    return {"avg": {k: (s / steps if steps else float("nan")) for k, s in zip(keys, sums)}, "steps": steps,
            "per_rank_steps": [int(round(v)) for v in mat[:, 0].tolist()],
            "per_rank_seconds": [round(v, 6) for v in mat[:, 1].tolist()],
            "scans_per_s": steps / slowest if slowest > 0 else float("nan")}


def write_synthetic_preprocessed(root, n, rank=0, world=1, n_points=24000, seed=0):
    """`n` synthetic preprocessed scans in the format preprocess_data.py:52-58 writes -- (n_points, 7) float64:
    xyz, unit normal, FDI-derived label 0..16 -- for runs without the dataset.  Rank r writes every world-th file."""
    from . import synth
    os.makedirs(root, exist_ok=True)
    for i in range(rank, n, world):
        path = os.path.join(root, f"SYN{i:05d}_{'upper' if i % 2 == 0 else 'lower'}_sampled_points.npy")
        if os.path.exists(path):
            continue
        cloud = synth.arch_cloud(n_points, seed=seed + i, with_normals=True).astype(np.float64)
        # labels: 16 teeth as angular sectors of the arch + gingiva (0) below the crown line -- any fixed function of position
        ang = np.arctan2(cloud[:, 1], cloud[:, 0])
        tooth = np.clip(((ang - ang.min()) / max(float(np.ptp(ang)), 1e-9) * 16).astype(np.int64), 0, 15) + 1
        tooth[cloud[:, 2] < np.median(cloud[:, 2])] = 0
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            np.save(f, np.concatenate([cloud[:, :6], tooth[:, None].astype(np.float64)], axis=1))
        os.replace(tmp, path)
