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
import time

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
        self._bufs, self._lists, self._recv, self._pending = [], [], [], None
        # Root only: where finish() spent its time, per finished gather -- waiting for the collective (what the render of the next
        # frame did not cover) and scattering the stripes into the frame.  GPU: event pairs on the unpack stream (read by timings());
        # CPU: host seconds.
        self._on_gpu = str(self.device).startswith("cuda")
        self._side = torch.cuda.Stream(device=self.device) if (self._on_gpu and rank == 0) else None   # the unpack runs beside the next render
        self._unpacked = {}            # slot -> event: its receive buffer has been read by the unpack
        self._sent = {}                # slot -> event: the collective that reads this slot's packed buffer (the root's own send buffer) is done
        self._marks = []               # (wait begin, wait end = unpack begin, unpack end)

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
            # ONE receive tensor per slot, [world][packed_bytes] (the gather's list = its rows): the unpack reads a whole group of ranks
            # with one strided copy instead of one copy per rank
            self._recv = [torch.empty((self.world, self.packed_bytes), dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
            self._lists = [[r[k] for k in range(self.world)] for r in self._recv]
        counts = [c for _, c in self.ranges]
        return [self._views(b, counts) for b in self._bufs]

    def alloc_full(self):
        """Root only: (fb [3][W*H] float32, u8 [W*H][3])."""
        fb = torch.zeros((3, self.npix), dtype=torch.float32, device=self.device)
        u8 = torch.zeros((self.npix, 3), dtype=torch.uint8, device=self.device)
        return fb, u8

    # ---- render + gather -----------------------------------------------------------------------------
    def begin(self, slot=None):
        """Before anything writes into packed buffer `slot` again (None: any slot): the current stream waits until the last gather
        that read it has completed.  Other ranks wait for their collective on the current stream in finish(); the root waits for it
        on its side stream only, so without this the render of frame k could overwrite the buffer the gather of frame k - slots is
        still sending from."""
        # A gather of this very slot that is still PENDING (gather_async without its finish(): slots=1, or a render into the slot
        # whose gather was the last one issued) has recorded no event yet -- finish it first, which does.
        if self._pending is not None and (slot is None or self._pending[1] == slot):
            self.finish()
        for k in ([slot] if slot is not None else list(self._sent)):
            ev = self._sent.pop(k, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def render(self, slot_views, spheres, render_fn, params=None, slot=None):
        """One launch per stripe of this rank, straight into the packed buffer's views (`slot`: which packed buffer they belong
        to; unknown = wait for every outstanding gather of this rank's buffers first)."""
        self.begin(slot)
        p = self.params if params is None else params
        for (b, c), (fb, u8) in zip(self.ranges, slot_views):
            render_fn(p, spheres, b, c, fb=fb, fb_u8=u8)

    def _unpack(self, recv, full_fb, full_u8):
        """recv [ranks][packed_bytes] -> the full frame.  Stripe s of ranks 0 .. world-1 are the consecutive parts s*world ..
        s*world + world-1 of the frame's world*stripes parts, i.e. ONE contiguous pixel run in which the first parts may be one pixel
        longer than the rest (split_range): at most two groups of equal-length parts per stripe, each moved by one strided copy for
        the float planes and one for the bytes (2 * stripes * (1 or 2) copies per frame; before round 4: 2 * stripes * world)."""
        ranks, ms = recv.shape[0], self.max_stripe
        parts = self.world * self.stripes
        base, extra = divmod(self.npix, parts)
        for s in range(self.stripes):
            p0 = s * self.world                                   # first part of this stripe's run
            groups = []
            if ranks != self.world:                               # a single local buffer (no process group): its own parts only
                groups = [(self.rank, 1)]
            else:
                n_long = max(0, min(self.world, extra - p0))      # parts p0 .. p0 + n_long - 1 have base + 1 pixels
                if n_long:
                    groups.append((0, n_long))
                if n_long < self.world:
                    groups.append((n_long, self.world - n_long))
            off = self.stripe_bytes * s
            for r0, n in groups:
                b, c = split_range(self.npix, p0 + r0, parts)
                rows = recv[0:1] if ranks != self.world else recv[r0:r0 + n]
                fb = rows[:, off:off + 12 * c].view(torch.float32).view(n, 3, c)
                u8 = rows[:, off + 12 * ms:off + 12 * ms + 3 * c].view(n, c, 3)
                full_fb[:, b:b + n * c].unflatten(1, (n, c)).copy_(fb.permute(1, 0, 2))
                full_u8[b:b + n * c].unflatten(0, (n, c)).copy_(u8)

    def pixel_counts(self):
        """Pixels every rank of the group renders (all ranks know the whole split)."""
        return [sum(c for _, c in stripe_ranges(self.npix, r, self.world, self.stripes)) for r in range(self.world)]

    def gather(self, slot=0, full_fb=None, full_u8=None):
        """ONE collective: every rank's packed buffer to rank 0 (the only root this class supports: the receive lists are
        allocated there), which scatters the stripes into the full framebuffer.  Without a process group (single process) it is a local copy; with one -- even of a
        single rank -- it goes through torch.distributed (RCCL on GPUs).  Blocking form of gather_async."""
        self.gather_async(slot, full_fb, full_u8)
        self.finish()
        self.drain()

    def gather_async(self, slot, full_fb=None, full_u8=None):
        """Enqueue the gather of slot `slot` (asynchronously: the collective runs on the backend's own
        stream after the work already queued on the current stream) and finish the previous one.  Shards may be
        unequal: every rank's buffer has the same padded size."""
        self.finish()
        if self.world == 1 and not dist.is_initialized():
            self._pending = ("local", slot, full_fb, full_u8, None)
            return
        if self.rank == 0:
            if slot in self._unpacked:       # the slot's receive buffer may still be read by the unpack of two frames ago (side stream)
                torch.cuda.current_stream().wait_event(self._unpacked.pop(slot))
            work = dist.gather(self._bufs[slot], self._lists[slot], dst=0, async_op=True)
        else:
            work = dist.gather(self._bufs[slot], None, dst=0, async_op=True)
        ready = None
        if self._side is not None:     # what the side stream must see before it unpacks: the frame tensors, made on the current stream
            ready = torch.cuda.Event()                            # BEFORE the next render is queued behind this call
            ready.record()
        self._pending = (work, slot, full_fb, full_u8, ready)

    def finish(self):
        """Wait for the pending gather (if any) and, on the root, scatter the stripes into the frame.  On a GPU the wait and the
        scatter run on a side stream (the next frame's render, already queued on the current stream, is not held up by them); the
        current stream joins the side stream in drain()."""
        if self._pending is None:
            return
        work, slot, full_fb, full_u8, ready = self._pending
        self._pending = None
        if work == "local":
            if full_fb is not None:
                self._unpack(self._bufs[slot].unsqueeze(0), full_fb, full_u8)
            return
        if self.rank != 0:
            work.wait()
            return
        if self._side is None:                                    # CPU (gloo): host clocks
            t0 = time.perf_counter()
            work.wait()
            t1 = time.perf_counter()
            if full_fb is not None:
                self._unpack(self._recv[slot], full_fb, full_u8)
            self._marks.append((t0, t1, time.perf_counter()))
            return
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        # The side stream waits for the current stream as it stood when the gather was ISSUED (the frame tensors exist), not as it
        # stands now: the next frame's render is usually queued already, and waiting for it would serialise the unpack behind it
        # and make gather_wait_ms read 0.
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            ev[0].record()
            work.wait()                                           # the side stream waits for the collective
            ev[1].record()
            self._sent[slot] = ev[1]                              # begin(slot): the current stream waits for it before the slot is rendered again
            if full_fb is not None:
                self._unpack(self._recv[slot], full_fb, full_u8)
            ev[2].record()
        self._unpacked[slot] = ev[2]
        self._marks.append(tuple(ev))

    def drain(self):
        """The current stream waits for everything finish() put on the side stream (call before reading the full frame)."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def timings(self):
        """Root: mean milliseconds per finished gather between (a) the side stream being free to look at it (the frame's own render
        done) and the collective's completion, and (b) scattering the stripes into the frame -> {"gather_wait_ms", "unpack_ms", "gathers"}; call after a device synchronise."""
        if not self._marks:
            return {"gather_wait_ms": None, "unpack_ms": None, "gathers": 0}
        if self._side is None:
            wait = [1e3 * (b - a) for a, b, _ in self._marks]
            unp = [1e3 * (c - b) for _, b, c in self._marks]
        else:
            wait = [a.elapsed_time(b) for a, b, _ in self._marks]
            unp = [b.elapsed_time(c) for _, b, c in self._marks]
        n = len(self._marks)
        return {"gather_wait_ms": round(sum(wait) / n, 4), "unpack_ms": round(sum(unp) / n, 4), "gathers": n}

    def reset_timings(self):
        self._marks = []


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
    shard.drain()
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
