"""How the hot path shards over the GPUs of one node (one process per GPU, torch.distributed over RCCL).

Rays are independent given read-only scene data, so there is no data-path collective; the only exchange is
the gather of finished framebuffers onto one GPU (AsyncGather: rank k % N for frame k). Two partitions are used:
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


def band_layout(rank: int, world: int, height: int, align: int = 8):
    """How one frame is cut into `world` row bands for rendering AND for the gather: (per_rows, (r0, r1), (s0, s1)).
    per_rows: common band height, a multiple of `align` (the 8-row ray packets); (r0, r1): the rows this rank renders -- shorter
    than per_rows for the last band that reaches the frame's end, EMPTY (r0 == r1) for ranks past it; (s0, s1): the slice this
    rank sends, in a render target padded to world * per_rows rows -- always per_rows rows, so every rank's send count is the
    same (a gather with unequal counts is undefined under NCCL/RCCL). Rank 0 keeps the first `height` rows of the assembly."""
    per = -(-height // world)
    per = -(-per // align) * align
    r0, r1 = min(height, rank * per), min(height, (rank + 1) * per)
    return per, (r0, r1), (rank * per, (rank + 1) * per)


def balanced_cuts(strip_cost, world: int, height: int, align: int = 8):
    """Row cuts [0, c1, ..., height] of `world` bands with about the same COST each, from the cost of every `align`-row strip of the
    frame (the per-tile cycle map of a whole-frame launch, summed over x: dust_hip_pipeline_tile_costs). Equal rows make bands whose
    launches differ by a third (the castle's top half 0.122 ms, its bottom half 0.096): the frame is done when the slowest is.
    Greedy prefix cuts at strip boundaries; a band may be empty when there are fewer strips than bands. Falls back to equal rows
    when the map is unusable."""
    import numpy as np
    n = -(-height // align)
    c = np.asarray(strip_cost, np.float64).reshape(-1) if strip_cost is not None else np.zeros(0)
    if len(c) != n or not np.isfinite(c).all() or c.sum() <= 0:
        per = -(-(-(-height // world)) // align) * align
        return [min(height, r * per) for r in range(world)] + [height]
    c = c + c.sum() * 1e-3 / n   # (strips nobody timed still take a moment)
    prefix = np.concatenate([[0.0], np.cumsum(c)])
    cuts = [0]
    for r in range(1, world):
        k = int(np.searchsorted(prefix, prefix[-1] * r / world))            # first strip boundary at or past the r-th share
        if k > 0 and prefix[-1] * r / world - prefix[k - 1] < prefix[k] - prefix[-1] * r / world:
            k -= 1                                                           # (the nearer boundary)
        cuts.append(min(height, max(cuts[-1], k * align)))
    return cuts + [height]


def rebalance_cuts(strip_cost, cuts, band_ms, world: int, height: int, align: int = 8):
    """One round of proportional correction of cost-balanced cuts from MEASURED band-step times (a frame is done when its slowest band
    is, and a tile-cost map taken from one whole-frame launch predicts a band's step under four frames in flight only to +-15 %: round 5's
    N = 8 emulation had seven bands at 0.032-0.034 ms and one at 0.0425). The strips of band r keep their relative costs and are scaled
    so that they add up to band_ms[r]; the cuts are then made again on the rescaled map. Returns (cuts, rescaled strip costs): feed the
    map back in for the next round. A band without rows or without a time keeps its strips' costs."""
    import numpy as np
    n = -(-height // align)
    c = np.asarray(strip_cost, np.float64).reshape(-1).copy() if strip_cost is not None else np.zeros(0)
    if len(c) != n or not np.isfinite(c).all() or c.sum() <= 0:
        c = np.ones(n)
    c = c / c.sum()
    c = c + 1e-3 / n
    out = c.copy()
    for r in range(world):
        lo, hi = cuts[r] // align, -(-cuts[r + 1] // align)
        t = float(band_ms[r]) if r < len(band_ms) else 0.0
        if hi > lo and t > 0.0 and np.isfinite(t):
            out[lo:hi] = c[lo:hi] * (t / c[lo:hi].sum())
    return balanced_cuts(out, world, height, align), out


def layout_from_cuts(rank: int, cuts, height: int, align: int = 8):
    """band_layout for bands of unequal height: (per_rows, (r0, r1), (s0, s1)) with per_rows the tallest band (a multiple of `align`)
    and the send slice [r0, r0 + per_rows) -- the band and whatever follows it in the rank's render target, which must therefore
    have at least max(r0) + per_rows rows. Every rank sends per_rows rows; the root keeps the first r1 - r0 of each."""
    per = max(cuts[r + 1] - cuts[r] for r in range(len(cuts) - 1))
    per = max(align, -(-per // align) * align)
    r0, r1 = cuts[rank], cuts[rank + 1]
    return per, (r0, r1), (r0, r0 + per)


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


def assemble_bands_from_cuts(parts, cuts):
    """Stack gathered bands of unequal height (layout_from_cuts: every part is per_rows tall, band r fills its first cuts[r+1] - cuts[r]
    rows) back into one frame."""
    import torch
    return torch.cat([p[: cuts[r + 1] - cuts[r]] for r, p in enumerate(parts)], dim=0)


class AsyncGather:
    """Double-buffered gather of finished frames: the gather of step k overlaps the rendering of step k+1.

    `slices=True` assembles by row slices instead: one all-to-all per step sends slice j of every rank's frame to rank j, so
    rank j ends up with slice j of all N frames (for spp sharding: every sample of its rows -- their sum needs no further
    traffic). Each rank then moves (N-1)/N of a frame over N-1 links AT ONCE, 2.1 MB per link and step for 1080p RGBA16F on 8 GPUs,
    where a gather makes the root's N-1 peers push a whole frame (16.6 MB) through one link each: ~0.3 ms at the ~55 GB/s a
    link sustains in one direction, longer than the frame takes to render. (Collectives of one communicator run one after the
    other, so gathers to different roots do not overlap either.) `last()` then returns the [N * slice_rows, ...] tensor.

    `rotate=True` gathers step k's frames to rank k % world instead of always to rank `dst`. xGMI is point to point: with a fixed
    root every peer pushes its whole frame through its ONE link to that root, step after step (16.6 MB per 1080p RGBA16F
    frame: ~0.3 ms at the ~55 GB/s a link sustains in one direction -- longer than the frame takes to render), while the other
    49 link directions of an 8-GPU node idle. Rotating the root spreads the same bytes over every link in both directions (each link then
    carries one frame every `world` steps); frame k is assembled on GPU k % world, which is also how an offline renderer would
    spread the encoding / writing of finished frames."""

    def __init__(self, dist, like, depth: int = 2, dst: int = 0, rotate: bool = False, slices: bool = False):
        import torch
        self.dist, self.dst, self.rotate, self.slices = dist, dst, rotate, slices
        self.active = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        self.bufs = [torch.empty_like(like) for _ in range(depth)]
        self.works = [None] * depth
        self.roots = [dst] * depth
        self.out = None
        if self.active and slices:
            assert like.shape[0] % self.world == 0, "slices: the frame's rows must divide by the world size (pad the target)"
            self.out = [torch.empty_like(like) for _ in range(depth)]
        elif self.active and (rotate or self.rank == dst):
            self.out = [[torch.empty_like(like) for _ in range(self.world)] for _ in range(depth)]
        self.k = 0

    def _root(self):
        return self.k % self.world if self.rotate else self.dst

    def _gather(self, b, tensor):
        if self.slices:  # one all-to-all: row slice j of every rank's frame lands on rank j
            self.roots[b] = self.rank
            self.works[b] = self.dist.all_to_all_single(self.out[b], tensor, async_op=True)
            return
        root = self._root()
        self.roots[b] = root
        self.works[b] = self.dist.gather(tensor, self.out[b] if (self.out is not None and self.rank == root) else None, dst=root,
                                         async_op=True)

    def submit(self, fill):
        """`fill(buf)` enqueues the copy of this rank's finished frame into `buf` on the current stream."""
        b = self.k % len(self.bufs)
        if self.works[b] is not None:
            self.works[b].wait()          # the send that last used this buffer is done (stream-ordered for NCCL)
        fill(self.bufs[b])
        if self.active:
            self._gather(b, self.bufs[b])
        self.k += 1
        return b

    def wait_slot(self, b):
        """Before slot b's source is rendered into again: the gather that last read it has completed."""
        b %= len(self.works)
        if self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None

    def submit_view(self, view):
        """Gather `view` (this rank's finished frame or band, still in its render target) without a staging copy. The caller
        alternates between len(self.bufs) targets and calls wait_slot(k) before rendering into one again."""
        b = self.k % len(self.bufs)
        self.wait_slot(b)
        self.bufs[b] = view
        if self.active:
            self._gather(b, view)
        self.k += 1
        return b

    def finish(self):
        for i, w in enumerate(self.works):
            if w is not None:
                w.wait()
                self.works[i] = None

    def last_root(self):
        """the rank the most recent submit gathered to"""
        return self.roots[(self.k - 1) % len(self.bufs)] if self.active else 0

    def last(self):
        """Frames gathered by the most recent submit (on its root only; [own buffer] when not distributed)."""
        b = (self.k - 1) % len(self.bufs)
        if not self.active:
            return [self.bufs[b]]
        if self.slices:
            return self.out[b]
        return self.out[b] if (self.out is not None and self.rank == self.roots[b]) else None


# ----------------------------------------------------------------------------------------------------------------------
# Multi-GPU diffuse GI (SURVEY 8e option i; the protocol is spelled out in include/dust_hip.h at
# dust_hip_pipeline_gi_exchange): pixel passes on row bands, identical hash + surfel pool on every GPU, the final
# gather's side effects exchanged with two small all-reduces and one all-gather per frame.

class DeviceArray:
    """A raw device pointer dressed as a CUDA-array-interface object so torch can alias it (no copy)."""

    def __init__(self, ptr: int, n_items: int, typestr: str = "<i4"):
        self.__cuda_array_interface__ = {"shape": (n_items,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def alias_exchange_buffers(ex, device="cuda"):
    """torch int32 views of the three buffers of a _lib.GiExchange: (slot_owner[pool], touched[rows*width], merged[pool*4])."""
    import torch
    owner = torch.as_tensor(DeviceArray(ex.slot_owner, ex.pool_size), device=device)
    touched = torch.as_tensor(DeviceArray(ex.touched, ex.touched_rows * ex.width), device=device)
    merged = torch.as_tensor(DeviceArray(ex.merged, ex.pool_size * 4), device=device)
    return owner, touched, merged


def gi_band_rows(world: int, height: int, align: int = 8) -> int:
    """Rows per band (the last band may be shorter); `touched` is allocated with world * gi_band_rows rows."""
    per = -(-height // world)
    return -(-per // align) * align


def gi_exchange_step(dist, rank, world, owner, touched, merged, band_items, export_fn, import_fn):
    """Steps 2-5 of the per-frame protocol. owner/touched/merged are int32 tensors on the exchange buffers (device
    tensors aliasing the library's buffers under RCCL; CPU tensors under gloo in the tests). band_items = items of
    `touched` per band (band_rows * width). export_fn / import_fn launch dust_hip_gi_export / dust_hip_gi_import
    (stream-ordered with the collectives: the context must use torch's current stream)."""
    import torch
    if dist is not None and dist.is_initialized() and world > 1:
        dist.all_reduce(owner, op=dist.ReduceOp.MAX)
        mine = touched[rank * band_items:(rank + 1) * band_items]
        if touched.is_cuda:
            dist.all_gather_into_tensor(touched, mine)          # in place: band r sits at offset r * band_items
        else:
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine.clone())
            touched.copy_(torch.cat(parts))
    export_fn()
    if dist is not None and dist.is_initialized() and world > 1:
        dist.all_reduce(merged, op=dist.ReduceOp.SUM)            # one contributor per slot, zeros elsewhere: exact
    import_fn()
