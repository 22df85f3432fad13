"""Multi-GPU: the path shards with no data-path exchange.  The reference already splits the
flat path range into 8 contiguous, independent blocks (src/render.cpp:9-10,24-27); here rank
r of R owns a contiguous range of x-major pixel indices (= a band of image columns), renders
it with the fused device path, and the framebuffer slices are collected on rank 0 by ONE
gather (RCCL over xGMI on GPUs: every peer has its own direct link to the root, so a direct
gather uses all links in parallel -- no ring).  One process per GPU; `torch.distributed` is
plumbing only (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).
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


class FrameShard:
    """Pixel range of one rank plus the packed buffer its slice travels in.

    Packed layout per rank (max_count = the largest shard, so every rank sends equal bytes):
        float32 [3][max_count]  clipped pixel values, planes r,g,b
        uint8   [max_count][3]  8-bit pixels
    """

    def __init__(self, params: RenderParams, rank=0, world=1, device=None, slots=1):
        """slots > 1 allocates that many packed buffers so that the gather of one frame can overlap
        the render of the next (see gather_async)."""
        self.params, self.rank, self.world, self.slots = params, rank, world, max(1, slots)
        self.npix = params.width * params.height
        self.pixel_begin, self.pixel_count = split_range(self.npix, rank, world)
        self.max_count = split_range(self.npix, 0, world)[1]
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self._packed = None
        self._gather_list = None
        self._slot_bufs = []
        self._pending = None

    @property
    def packed_bytes(self):
        return 15 * self.max_count

    def alloc(self):
        """-> (fb [3][pixel_count] float32, u8 [pixel_count][3]) views into the packed buffer.
        The fb view is [3][max_count] storage with the first pixel_count columns used when the
        shard is smaller than max_count, so kernels get a dense [3][pixel_count] only when
        pixel_count == max_count; otherwise a private dense buffer is used and packed later."""
        self._packed = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=self.device)
        if self.pixel_count == self.max_count:
            fb = self._packed[:12 * self.max_count].view(torch.float32).view(3, self.max_count)
            u8 = self._packed[12 * self.max_count:].view(self.max_count, 3)
            self._dense = None
        else:
            fb = torch.zeros((3, self.pixel_count), dtype=torch.float32, device=self.device)
            u8 = torch.zeros((self.pixel_count, 3), dtype=torch.uint8, device=self.device)
            self._dense = (fb, u8)
        return fb, u8

    def alloc_full(self):
        """Root only: (fb [3][W*H] float32, u8 [W*H][3])."""
        fb = torch.zeros((3, self.npix), dtype=torch.float32, device=self.device)
        u8 = torch.zeros((self.npix, 3), dtype=torch.uint8, device=self.device)
        return fb, u8

    def _pack(self):
        if self._dense is not None:
            fb, u8 = self._dense
            pf = self._packed[:12 * self.max_count].view(torch.float32).view(3, self.max_count)
            pf[:, :self.pixel_count] = fb
            self._packed[12 * self.max_count:].view(self.max_count, 3)[:self.pixel_count] = u8

    def gather(self, fb, u8, full_fb=None, full_u8=None, dst=0):
        """ONE collective: every rank's packed slice to `dst`, which scatters the slices into
        the full framebuffer.  Without a process group (single process) it is a local copy; with
        one -- even of a single rank -- it goes through torch.distributed (RCCL on GPUs)."""
        if self.world == 1 and not dist.is_initialized():
            if full_fb is not None:
                full_fb.copy_(fb)
                full_u8.copy_(u8)
            return
        self._pack()
        if self.rank == dst:
            if self._gather_list is None:
                self._gather_list = [torch.empty_like(self._packed) for _ in range(self.world)]
            dist.gather(self._packed, self._gather_list, dst=dst)
            for r, buf in enumerate(self._gather_list):
                b, c = split_range(self.npix, r, self.world)
                full_fb[:, b:b + c] = buf[:12 * self.max_count].view(torch.float32).view(3, self.max_count)[:, :c]
                full_u8[b:b + c] = buf[12 * self.max_count:].view(self.max_count, 3)[:c]
        else:
            dist.gather(self._packed, None, dst=dst)


    # ---- pipelined form: the gather of frame k overlaps the render of frame k+1 -------------------
    def alloc_slots(self):
        """-> list of (fb, u8) view pairs, one per slot (requires equal shards: pixel_count == max_count)."""
        if self.pixel_count != self.max_count:
            raise ValueError("pipelined gather needs equal shards")
        self._slot_bufs = [torch.zeros(self.packed_bytes, dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        if self.rank == 0:
            self._slot_lists = [[torch.empty_like(self._slot_bufs[0]) for _ in range(self.world)] for _ in range(self.slots)]
        return [(b[:12 * self.max_count].view(torch.float32).view(3, self.max_count),
                 b[12 * self.max_count:].view(self.max_count, 3)) for b in self._slot_bufs]

    def gather_async(self, slot, full_fb=None, full_u8=None, dst=0):
        """Enqueue the gather of slot `slot` (asynchronously: the collective runs on the backend's own
        stream after the work already queued on the current stream) and finish the previous one."""
        self.finish()
        if self.world == 1 and not dist.is_initialized():
            self._pending = ("local", slot, full_fb, full_u8)
            return
        if self.rank == dst:
            work = dist.gather(self._slot_bufs[slot], self._slot_lists[slot], dst=dst, async_op=True)
        else:
            work = dist.gather(self._slot_bufs[slot], None, dst=dst, async_op=True)
        self._pending = (work, slot, full_fb, full_u8)

    def finish(self):
        """Wait for the pending gather (if any) and, on the root, scatter the slices into the frame."""
        if self._pending is None:
            return
        work, slot, full_fb, full_u8 = self._pending
        self._pending = None
        if work == "local":
            bufs = [self._slot_bufs[slot]]
        else:
            work.wait()
            if self.rank != 0:
                return
            bufs = self._slot_lists[slot]
        if full_fb is not None:
            for r, buf in enumerate(bufs):
                b, c = split_range(self.npix, r, self.world)
                full_fb[:, b:b + c] = buf[:12 * self.max_count].view(torch.float32).view(3, self.max_count)[:, :c]
                full_u8[b:b + c] = buf[12 * self.max_count:].view(self.max_count, 3)[:c]


def render_frame_sharded(params: RenderParams, spheres, rank=None, world=None, render_fn=None, device=None):
    """Render this rank's pixel band and gather the image on rank 0.
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
    shard = FrameShard(params, rank, world, device=device)
    fb, u8 = shard.alloc()
    render_fn(params, spheres, shard.pixel_begin, shard.pixel_count, fb=fb, fb_u8=u8)
    full = shard.alloc_full() if rank == 0 else (None, None)
    shard.gather(fb, u8, *full)
    return full


def recorded_traffic(root):
    """HBM bytes per launch of the headline kernel from the committed PMC profile
    (profiles/hbm_traffic.json, written from separate rocprofv3 --pmc passes), or None."""
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch")
