"""Sample sharding across ranks: one process per GPU, one all-reduce per frame.

Samples of a pixel are independent, so the global sample range [0, spp) is cut into one contiguous
block per rank (the remainder goes to the low ranks).  Random numbers are keyed by the GLOBAL sample
index, so the set of samples in a frame does not depend on the number of ranks.  Every rank adds its
samples, unnormalised, into a full-frame float buffer; the buffers are summed with one all-reduce
(RCCL over xGMI under torch.distributed's "nccl" backend; gloo in the CPU tests) and scaled by 1/spp,
the `acc * sppInv` of the reference's Renderer::doWork (src/Renderer.cpp:102-105).
"""


def sample_range(spp, world, rank):
    """(first_sample, count) of `rank`'s block of the global sample range."""
    if not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank / world / spp")
    base, rem = divmod(spp, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def reduce_frame(frame_sum, total_spp, dist=None):
    """All-reduces the per-rank radiance sums in place and returns the mean image (a view of the
    same tensor scaled in place)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(frame_sum)
    frame_sum.mul_(1.0 / float(total_spp))
    return frame_sum
