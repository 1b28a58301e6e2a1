"""K-sharding over the GPUs of one node: who owns which samples, and the single
exchange step of an iteration.  One process per GPU, ``torch.distributed`` ("nccl" = RCCL over xGMI on
GPUs; "gloo" for CPU rehearsals)."""
from __future__ import annotations


def shard_range(K: int, rank: int, world: int):
    """Contiguous shard [k0, k0 + n) of the K global samples owned by ``rank`` (first K % world ranks get
    one extra).  The exploit/explore split and the Philox counter use the GLOBAL index, so the union of
    the shards reproduces the unsharded draw."""
    base, rem = divmod(int(K), int(world))
    n = base + (1 if rank < rem else 0)
    return rank * base + min(rank, rem), n


def exchange_partials(partial, world: int, group=None, out=None):
    """All-gather the per-rank softmin record {rho, eta, eta2, W[T,2]} (3+2T doubles): the ONE collective
    of an iteration.  ``partial`` is a 1-D float64 tensor (CUDA under nccl, CPU or CUDA under gloo);
    returns the [world * n] concatenation on the same device."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = partial.new_empty(world * partial.numel())
    backend = dist.get_backend(group)
    if backend == "nccl" or not partial.is_cuda:
        if backend == "nccl":
            dist.all_gather_into_tensor(out, partial, group=group)
        else:
            chunks = list(out.view(world, -1).unbind(0))
            dist.all_gather(chunks, partial, group=group)
    else:  # CUDA tensors over gloo: stage through host memory
        host = partial.cpu()
        chunks = [host.new_empty(host.shape) for _ in range(world)]
        dist.all_gather(chunks, host, group=group)
        out.copy_(torch.cat(chunks))
    return out
