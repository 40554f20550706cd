"""The Python-side switches of the package, in ONE object.

Rounds 1-4 read them from the environment into module-level globals at import time (pointnet2_utils.FUSED_SA,
pointops.KNN_GRID, _lib.INDEX_CHECK, ...).  They now live in ``config.cfg``; the operators read it at call time, the
``TGN_*`` environment variables only seed it once, when this module is first imported, and

    with config.override(fused_sa=False, knn_cache_size=0):
        ...

changes them for a block.  The old module attributes still work (reads and writes are forwarded to ``cfg``: see
``legacy_attributes``), so scripts written against rounds 1-4 keep their meaning.  The KERNEL-side switches are a
separate table inside the library (``_lib.set_tuning`` / ``tgn_set_tuning``, include/tgn_pointops.h).

field                 env (seed only)        meaning
fused_sa              TGN_FUSED_SA=0/1       eval-mode set abstraction / feature propagation on the fused kernel paths
commute_fp            TGN_COMMUTE_FP=0/1     feature propagation: first 1x1 convolution on the coarse points (exact algebra)
sa_bf16x3             TGN_SA_BF16X3=0/1      second layer of the chained set-abstraction kernel as six bf16 MFMAs per fp32
                                             product (fp32-class error on FINITE inputs; an infinite or > 3.39e38 activation or
                                             weight turns into NaN there, where the exact-fp32 form (0) gives inf)
knn_grid              TGN_KNN_GRID=0/1       kNN through the per-segment grid kernel where segments are large enough
knn_grid_min_points   TGN_KNN_GRID_MIN       ... "large enough": average points per segment (3000)
knn_cache_size        TGN_KNN_CACHE          entries of the kNN memo (blocks of one stage share their neighbour lists); 0 = off
index_check           TGN_INDEX_CHECK        "sync": gather operators read their stream's index-error flag after the launch and
                                             raise IndexError like torch's advanced indexing; "off": no check, no synchronisation
"""
import dataclasses
import os
import sys
import types


def _flag(name, default="1"):
    return os.environ.get(name, default) != "0"


@dataclasses.dataclass
class Config:
    fused_sa: bool = True
    commute_fp: bool = True
    sa_bf16x3: bool = True
    knn_grid: bool = True
    knn_grid_min_points: int = 3000
    knn_cache_size: int = 16
    index_check: str = "sync"

    @classmethod
    def from_env(cls):
        return cls(fused_sa=_flag("TGN_FUSED_SA"), commute_fp=_flag("TGN_COMMUTE_FP"), sa_bf16x3=_flag("TGN_SA_BF16X3"),
                   knn_grid=_flag("TGN_KNN_GRID"), knn_grid_min_points=int(os.environ.get("TGN_KNN_GRID_MIN", "3000")),
                   knn_cache_size=int(os.environ.get("TGN_KNN_CACHE", "16")),
                   index_check=os.environ.get("TGN_INDEX_CHECK", "sync").lower())


cfg = Config.from_env()


class override:
    """``with override(field=value, ...):`` -- set fields of ``cfg`` for a block (unknown fields raise)."""

    def __init__(self, **fields):
        for k in fields:
            if k not in Config.__dataclass_fields__:
                raise AttributeError(f"toothgroupnetwork_amd.config has no switch {k!r}")
        self.fields, self.prev = fields, {}

    def __enter__(self):
        for k, v in self.fields.items():
            self.prev[k] = getattr(cfg, k)
            setattr(cfg, k, v)
        return cfg

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            setattr(cfg, k, v)
        return False


def legacy_attributes(module_name, mapping):
    """Make ``module.OLD_NAME`` a live alias of ``cfg.<field>`` (reads and writes) for the module-level switch names of rounds
    1-4: the module's class is replaced by a subclass carrying one property per name."""
    mod = sys.modules[module_name]

    def alias(field):
        return property(lambda self: getattr(cfg, field), lambda self, value: setattr(cfg, field, value))

    mod.__class__ = type("_ModuleWithSwitches", (types.ModuleType,), {old: alias(field) for old, field in mapping.items()})
