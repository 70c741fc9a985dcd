"""Body of tests/test_sharding_gloo.py::test_rccl_arms_at_world_one (GPU): the exchange helpers of coral_amd.sharding with backend
"nccl" (= RCCL) and device tensors, in a world of ONE rank — the only RCCL world a one-GPU box can form (two ranks cannot share a
device under RCCL).  It proves the calls, dtypes and devices of the RCCL arms (all_gather of counts and padded rows, all_reduce,
gather to rank 0, broadcast of the command header + payload); the multi-rank semantics are what the gloo tests check."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from coral_amd import sharding
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))

    class DR:
        rank, world, group, device = 0, 1, None, torch.device("cuda:0")
    dr = DR()
    assert dist.get_backend() == "nccl" and sharding._comm_device(dr).type == "cuda"
    rows = torch.arange(21, dtype=torch.int64, device="cuda:0").reshape(7, 3)
    got = sharding.allgather_rows(dr, rows)
    assert got.is_cuda and torch.equal(got, rows)
    assert sharding.allgather_rows(dr, rows[:0]).shape == (0, 3)
    t = torch.tensor([[3, 4], [5, 6]], dtype=torch.int64, device="cuda:0")
    assert torch.equal(sharding.allreduce_sum(dr, t.clone()), t)
    buf = np.arange(1000, dtype=np.uint8)
    back = sharding._gather_to_rank0(dr, buf)
    assert len(back) == 1 and np.array_equal(back[0], buf)
    sharding.command(dr, sharding.CMD_COVERAGE, payload=np.array([[1, 2, 3], [4, 5, 6]], dtype=np.int64))      # broadcasts from rank 0 (to nobody)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl world-1 ok")


if __name__ == "__main__":
    main()
