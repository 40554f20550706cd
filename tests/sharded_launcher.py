"""Test-side launcher of tools/preprocess_sharded.py for boxes without a GPU: runs the runner's own main() with the
ORACLE's FPS injected as the sampler, so that the control flow (sharding, per-rank files, the one all_gather) can be
exercised on gloo ranks.  The runner itself knows nothing about the oracle (tests/test_capi_symbols.py checks)."""
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def oracle_fps_batch(xyz_list, npoint):
    from oracle import cpu as O
    return [O.furthestsampling(np.ascontiguousarray(x, dtype=np.float32), [x.shape[0]], [npoint]).reshape(-1) for x in xyz_list]


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("preprocess_sharded", os.path.join(REPO, "tools", "preprocess_sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(sys.argv[1:], fps_batch=oracle_fps_batch, script=os.path.abspath(__file__))
