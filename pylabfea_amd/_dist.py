"""Host-staged transport of the library's collectives over a ``torch.distributed`` process group (gloo on CPU, or any
backend): what ``Model.distribute(..., uid=None, host_allreduce=fn)`` / ``plfx_comm_init_callback`` expect.  Used where RCCL
cannot be (several ranks on ONE GPU in the tests, hosts without RCCL); the product transport is RCCL inside the library."""
import numpy as np

HALO_EXCHANGE = 100   # op code of the neighbour exchange (include/plfx.h: plfx_allreduce_fn)


def host_transport(dist, rank, world):
    """fn(array, op): in-place all-reduce of a NumPy array (op 0 = sum, 3 = min) or, for op 100, the halo exchange of a
    strip: ``array`` holds [slab for the left neighbour | slab for the right neighbour] and comes back as [slab from the
    left neighbour | slab from the right neighbour] (zeros where there is no neighbour)."""
    import torch

    def fn(arr, op):
        t = torch.from_numpy(arr)           # shares memory with the library's staging buffer
        if op == HALO_EXCHANGE:
            n = arr.size // 2
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            left = outs[rank - 1][n:].numpy() if rank > 0 else np.zeros(n)          # its slab "for the right neighbour"
            right = outs[rank + 1][:n].numpy() if rank < world - 1 else np.zeros(n)  # its slab "for the left neighbour"
            arr[:n] = left
            arr[n:] = right
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN if op == 3 else dist.ReduceOp.SUM)
    return fn
