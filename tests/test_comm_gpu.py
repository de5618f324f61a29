"""Native RCCL all-gather of costs (anet_comm_*) with a single rank: the code path bench-independent
C/C++ hosts use for multi-GPU.  (2/4/8-rank runs need as many GPUs; the box has one.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_native_allgather_single_rank(anet_ctx):
    import torch
    import allocnet_amd as aa
    from allocnet_amd.distributed import NativeComm
    ctx = aa.Context(0)              # own context: the communicator lives in it
    comm = NativeComm(ctx, 1, 0)
    assert len(comm.unique_id) == 128
    send = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    recv = torch.full((1000,), -1.0, dtype=torch.float64, device="cuda")
    comm.allgather_costs(send, recv, 1000)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    with pytest.raises(aa.AnetError):            # second init on the same context is refused
        NativeComm(ctx, 1, 0)
    comm.close()
    ctx.close()
