"""CPU, world_size 2, gloo: the N>1 path of the sharded runner (one collective: the final metric gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from toothgroupnetwork_amd import sharding
    r, lr, w, device = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    items = list(range(11))
    res = sharding.run_sharded(items, lambda i: {"sum": float(i), "one": 1.0}, rank, world, mode="round_robin")
    mat = sharding.gather_metrics([float(rank), 10.0 + rank])
    sharding.barrier()
    mx = sharding.max_over_ranks(1.0 + rank)
    q.put((rank, res, mat.tolist(), mx))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_sharded_run_and_metric_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, mat, mx in outs:
        assert res["count"] == 11 and res["sum"] == 55.0 and res["one"] == 11.0
        assert res["per_rank_count"] == [6, 5]
        assert mat == [[0.0, 10.0], [1.0, 11.0]]
        assert mx == float(world)
