"""Multi-GPU: the path shards with no data-path exchange.  The reference already splits the
flat path range into 8 contiguous, independent blocks (src/render.cpp:9-10,24-27); here rank
r of R owns contiguous ranges of x-major pixel indices (= bands of image columns), renders
them with the fused device path, and the framebuffer slices are collected on rank 0 by ONE
gather (RCCL over xGMI on GPUs: every peer has its own direct link to the root, so a direct
gather uses all links in parallel -- no ring).  One process per GPU; `torch.distributed` is
plumbing only (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

Sharding.  The frame's pixel range is cut into `world * stripes` contiguous stripes (near-equal,
the first `npix % (world*stripes)` one pixel longer); rank r renders stripes r, r + world,
r + 2*world, ...  stripes = 1 is the reference's split (one contiguous band per rank); stripes > 1
interleaves the bands, which evens out APT_FLAG_RETIRE's column-dependent cost (the mirror-ball
columns retire differently) at the price of one launch per stripe.

Packed buffer per rank = its stripes back to back, every stripe in a slot sized for the LARGEST
stripe, so that all ranks send equal bytes whatever the split:
    per stripe:  float32 region of 3*max_stripe floats holding the dense [3][count] planes at its start,
                 uint8   region of 3*max_stripe bytes  holding the dense [count][3] pixels at its start.
The kernels write straight into these regions (no pack step); the root unpacks with the true counts.
"""
import json
import os

import torch
import torch.distributed as dist

from ._lib import RenderParams


def split_range(total, rank, world):
    """Contiguous near-equal split of [0,total): -> (begin, count).  The first total%world
    ranks get one extra element."""
    base, extra = divmod(total, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)


def stripe_ranges(npix, rank, world, stripes=1):
    """The (begin, count) pixel ranges rank `rank` renders: stripes rank, rank+world, ... of world*stripes."""
    parts = world * stripes
    return [split_range(npix, s * world + rank, parts) for s in range(stripes)]


class FrameShard:
    """The pixel ranges of one rank plus the packed buffer(s) its slices travel in."""

    def __init__(self, params: RenderParams, rank=0, world=1, device=None, slots=1, stripes=1):
        """slots > 1 allocates that many packed buffers so that the gather of one frame can overlap
        the render of the next (see gather_async)."""
        self.params, self.rank, self.world = params, rank, world
        self.slots, self.stripes = max(1, slots), max(1, stripes)
        self.npix = params.width * params.height
        if world * self.stripes > self.npix:
            raise ValueError("more stripes than pixels")
        self.ranges = stripe_ranges(self.npix, rank, world, self.stripes)
        self.max_stripe = split_range(self.npix, 0, world * self.stripes)[1]
        self.pixel_count = sum(c for _, c in self.ranges)          # pixels this rank renders
        self.pixel_begin = self.ranges[0][0]                         # (meaningful as a range only for stripes == 1)
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self._bufs, self._lists, self._pending = [], [], None

    # ---- layout ------------------------------------------------------------------------------------
    @property
    def stripe_bytes(self):
        return (15 * self.max_stripe + 15) // 16 * 16      # 16-byte aligned stripe slots (float views, vector stores)

    @property
    def packed_bytes(self):
        return self.stripe_bytes * self.stripes

    def _views(self, buf, counts):
        """-> [(fb [3][count] float32, u8 [count][3]) per stripe] as views into a packed buffer."""
        out, ms = [], self.max_stripe
        for s, c in enumerate(counts):
            base = self.stripe_bytes * s
            fb = buf[base:base + 12 * c].view(torch.float32).view(3, c)
            u8 = buf[base + 12 * ms:base + 12 * ms + 3 * c].view(c, 3)
            out.append((fb, u8))
        return out

    def alloc_slots(self):
        """Allocates the packed buffers; -> per slot the list of (fb, u8) views, one pair per stripe."""
        self._bufs = [torch.zeros(self.packed_bytes, dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        if self.rank == 0:
            self._lists = [[torch.empty_like(self._bufs[0]) for _ in range(self.world)] for _ in range(self.slots)]
        counts = [c for _, c in self.ranges]
        return [self._views(b, counts) for b in self._bufs]

    def alloc_full(self):
        """Root only: (fb [3][W*H] float32, u8 [W*H][3])."""
        fb = torch.zeros((3, self.npix), dtype=torch.float32, device=self.device)
        u8 = torch.zeros((self.npix, 3), dtype=torch.uint8, device=self.device)
        return fb, u8

    # ---- render + gather -----------------------------------------------------------------------------
    def render(self, slot_views, spheres, render_fn, params=None):
        """One launch per stripe of this rank, straight into the packed buffer's views."""
        p = self.params if params is None else params
        for (b, c), (fb, u8) in zip(self.ranges, slot_views):
            render_fn(p, spheres, b, c, fb=fb, fb_u8=u8)

    def _unpack(self, bufs, full_fb, full_u8):
        for r, buf in enumerate(bufs):
            ranges = stripe_ranges(self.npix, r, self.world, self.stripes)
            for (b, c), (fb, u8) in zip(ranges, self._views(buf, [c for _, c in ranges])):
                full_fb[:, b:b + c] = fb
                full_u8[b:b + c] = u8

    def pixel_counts(self):
        """Pixels every rank of the group renders (all ranks know the whole split)."""
        return [sum(c for _, c in stripe_ranges(self.npix, r, self.world, self.stripes)) for r in range(self.world)]

    def gather(self, slot=0, full_fb=None, full_u8=None):
        """ONE collective: every rank's packed buffer to rank 0 (the only root this class supports: the receive lists are
        allocated there), which scatters the stripes into the full framebuffer.  Without a process group (single process) it is a local copy; with one -- even of a
        single rank -- it goes through torch.distributed (RCCL on GPUs).  Blocking form of gather_async."""
        self.gather_async(slot, full_fb, full_u8)
        self.finish()

    def gather_async(self, slot, full_fb=None, full_u8=None):
        """Enqueue the gather of slot `slot` (asynchronously: the collective runs on the backend's own
        stream after the work already queued on the current stream) and finish the previous one.  Shards may be
        unequal: every rank's buffer has the same padded size."""
        self.finish()
        if self.world == 1 and not dist.is_initialized():
            self._pending = ("local", slot, full_fb, full_u8)
            return
        if self.rank == 0:
            work = dist.gather(self._bufs[slot], self._lists[slot], dst=0, async_op=True)
        else:
            work = dist.gather(self._bufs[slot], None, dst=0, async_op=True)
        self._pending = (work, slot, full_fb, full_u8)

    def finish(self):
        """Wait for the pending gather (if any) and, on the root, scatter the stripes into the frame."""
        if self._pending is None:
            return
        work, slot, full_fb, full_u8 = self._pending
        self._pending = None
        if work == "local":
            bufs = [self._bufs[slot]]
        else:
            work.wait()
            if self.rank != 0:
                return
            bufs = self._lists[slot]
        if full_fb is not None:
            self._unpack(bufs, full_fb, full_u8)


def render_frame_sharded(params: RenderParams, spheres, rank=None, world=None, render_fn=None, device=None, stripes=1):
    """Render this rank's stripes and gather the image on rank 0.
    -> (fb [3][W*H], u8 [W*H][3]) on rank 0, (None, None) elsewhere.
    `render_fn(params, spheres, pixel_begin, pixel_count, fb=..., fb_u8=...)` defaults to the HIP
    path (render.render_frame, which raises without a GPU)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if render_fn is None:
        from . import render
        render_fn = render.render_frame
    shard = FrameShard(params, rank, world, device=device, stripes=stripes)
    views = shard.alloc_slots()[0]
    shard.render(views, spheres, render_fn)
    full = shard.alloc_full() if rank == 0 else (None, None)
    shard.gather(0, *full)
    return full


def recorded_traffic(root):
    """HBM bytes per launch of the headline kernel from the committed PMC profile (profiles/hbm_traffic.json,
    written from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes by profiles/summarize.py) together
    with the build tag it was recorded for -> (bytes or None, tag or None).  bench.py cannot collect PMC counters
    on itself (rocprofv3 has to wrap the process), so the file carries the id of the build it was recorded on
    (_lib.build_id(): a hash of the library's sources) and the figure is only reported for that very build:
    any other build gets (None, "recorded for build <id>, running <id>")."""
    from . import _lib
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        d = json.load(f)
    have, want = d.get("build_id"), _lib.build_id()
    if have != want:
        return None, f"profiles/hbm_traffic.json was recorded for build {have}, this is build {want}"
    return d.get("hbm_bytes_per_launch"), d.get("tag")
