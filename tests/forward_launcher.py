"""Test-side launcher of tools/forward_sharded.py for boxes without a GPU: runs the runner's own main() with a CPU step
object (a fixed function of each scan's arrays) in place of the GPU networks, so that the control flow -- self-spawning
the ranks, round-robin shards, the LossMeter sums, the one all_gather -- runs on gloo ranks."""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class CpuStep:
    keys = ("tooth_class_loss_1_val", "aux_val", "total_val")

    def __init__(self, device):
        self.device = device

    def __call__(self, batch_idx, batch_item):
        from toothgroupnetwork_amd import eval_sharded
        feat, seg = batch_item["feat"], batch_item["gt_seg_label"]
        assert feat.shape[:2] == (1, 6) and seg.shape[:2] == (1, 1) and int(seg.min()) >= -1
        a = float(feat.double().abs().mean()) + batch_idx
        b = float((seg + 1).double().mean())
        return eval_sharded.print_dict({"tooth_class_loss_1": (a, 1), "aux": (b, 0.5)}, "val")


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("forward_sharded", os.path.join(REPO, "tools", "forward_sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(sys.argv[1:], step_factory=CpuStep, script=os.path.abspath(__file__))
