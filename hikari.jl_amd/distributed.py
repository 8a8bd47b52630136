"""Multi-GPU sharding of the VolPath path: units are (pixel, sample_idx) paths with no
cross-path communication, so each rank renders the sample indices  rank+1, rank+1+G, ...  for ALL pixels
(ZSobol is a pure function of (px, py, sample_idx, dim), sampler/sobol.jl:269-309) and ONE sum-reduce of the
film accumulators [pixel_rgb 3N | pixel_weight_sum N] to rank 0 finishes the frame.  torch.distributed is
plumbing only (backend "nccl" is RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""


def shard_samples(total_samples, rank, world):
    """-> (first_sample_idx (1-based), count, stride) for this rank."""
    count = (total_samples - rank + world - 1) // world if total_samples > rank else 0
    return rank + 1, count, world


def shard_tiles(width, height, rank, world):
    """Pixel-tile sharding (the other partition of SURVEY 8e): `world` horizontal bands whose heights are multiples of the 8-row
    path tile (so no band splits a wave's 8x8 pixel block) -> (x0, y0, x1, y1) of this rank; bands of the last ranks may be empty."""
    rows8 = (height + 7) // 8
    lo = (rows8 * rank) // world * 8
    hi = (rows8 * (rank + 1)) // world * 8
    return 0, min(lo, height), width, min(hi, height)


def reduce_film(accum_tensor, root=0):
    """Sum-reduce the film accumulators onto `root` (16 MiB at 1024^2: one small collective per frame)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and accum_tensor.is_cuda:   # CPU-side test configuration: stage through the host
            host = accum_tensor.cpu()
            dist.reduce(host, dst=root, op=dist.ReduceOp.SUM)
            accum_tensor.copy_(host)
        else:
            dist.reduce(accum_tensor, dst=root, op=dist.ReduceOp.SUM)
    return accum_tensor
