"""How the hot path shards over the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Rays are independent given read-only scene data, so there is no data-path collective; the only exchange is
the gather of finished framebuffers to rank 0. Two partitions are used:
  * row bands of one frame  (DustHipFrameParams.row_begin/row_end)
  * samples of one view     ("N spp" = N consecutive frame indices) -- what bench.py scales with
Both are exercised on CPU with the gloo backend in tests/test_distributed_cpu.py.
"""


def band_rows(rank: int, world: int, height: int, align: int = 8):
    """Contiguous row band of `rank`, multiples of `align` rows (the 8x8 ray packets) except the last."""
    per = -(-height // world)
    per = -(-per // align) * align
    r0 = min(height, rank * per)
    r1 = min(height, r0 + per)
    return r0, r1


def sample_frame_index(step: int, rank: int, world: int, first: int = 1) -> int:
    """frame_index of the sample rank `rank` renders in step `step` (frame_index starts at 1, standard.rs:252)."""
    return first + step * world + rank


def gather_to_root(dist, local, dst: int = 0):
    """Gather equally-shaped per-rank tensors to `dst`. Returns the list on dst, None elsewhere.
    With world size 1 (dist is None) returns [local]."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    import torch
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())] if dist.get_rank() == dst else None
    dist.gather(local, out, dst=dst)
    return out


def assemble_bands(parts, height: int, align: int = 8):
    """Stack gathered row bands (each padded to the common band height) back into one frame."""
    import torch
    world = len(parts)
    rows = []
    for r, p in enumerate(parts):
        r0, r1 = band_rows(r, world, height, align)
        rows.append(p[: r1 - r0])
    return torch.cat(rows, dim=0)


class AsyncGather:
    """Double-buffered gather to rank 0: the gather of step k overlaps the rendering of step k+1.
    `fill(buf)` must enqueue the copy of this rank's finished frame into `buf` on the current stream."""

    def __init__(self, dist, like, depth: int = 2, dst: int = 0):
        import torch
        self.dist, self.dst = dist, dst
        self.active = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        self.bufs = [torch.empty_like(like) for _ in range(depth)]
        self.works = [None] * depth
        self.out = None
        if self.active and dist.get_rank() == dst:
            self.out = [[torch.empty_like(like) for _ in range(dist.get_world_size())] for _ in range(depth)]
        self.k = 0

    def submit(self, fill):
        b = self.k % len(self.bufs)
        if self.works[b] is not None:
            self.works[b].wait()          # the send that last used this buffer is done (stream-ordered for NCCL)
        fill(self.bufs[b])
        if self.active:
            self.works[b] = self.dist.gather(self.bufs[b], self.out[b] if self.out is not None else None, dst=self.dst,
                                             async_op=True)
        self.k += 1
        return b

    def finish(self):
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None

    def last(self):
        """Frames gathered by the most recent submit (rank dst only; [own buffer] when not distributed)."""
        b = (self.k - 1) % len(self.bufs)
        return self.out[b] if self.out is not None else [self.bufs[b]]
