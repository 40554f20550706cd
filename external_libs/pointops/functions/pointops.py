"""Drop-in for the reference's ``external_libs/pointops/functions/pointops.py``.

With ``sys.path = [<this repo>, <reference checkout>]`` the reference's models
(``models/modules/cbl_point_transformer/blocks.py:6``, ``heads.py:6``, ``basic_operators.py:4``,
``gen_utils.py:8``) import THIS module; the operators are implemented in
``toothgroupnetwork_amd.pointops`` on top of libtgn_pointops.so (HIP, gfx950).
``external_libs/`` deliberately has no ``__init__.py`` (namespace package), so
``external_libs.scheduler`` still resolves to the reference's own copy.
"""
from toothgroupnetwork_amd.pointops import (  # noqa: F401
    Aggregation,
    FurthestSampling,
    Grouping,
    Interpolation,
    KNNQuery,
    Subtraction,
    aggregation,
    furthestsampling,
    grouping,
    interpolation,
    interpolation2,
    knnquery,
    queryandgroup,
    subtraction,
)
