"""Sort-last compositing across ranks.

The reference has no distributed code: its only decomposition is MetaVolume::brick's spatial
grid drawn serially in visibility order on one GPU (MetaVolume.cpp:1369-1452,
NV20VolRen3D.cpp:190-231).  Here each rank owns a convex union of bricks (smk_set_shard),
ray-marches the full viewport against it, and the partial premultiplied-RGBA images are merged
with ONE exchange step:

    direct-send all-to-all of 1/P image tiles  ->  ordered "over" of P layers  ->  gather

"over" is associative but not commutative, so this cannot be an all-reduce(sum); the layer
order is the BSP front-to-back order of the shards for the current eye point.  On MI355X
`backend="nccl"` is RCCL over xGMI; the all-to-all uses all 7 links of each GPU at once.
The compositor is a parameter: the product passes the HIP kernel (smk_composite_over_device);
the gloo tests on CPU pass their own checker, so the same plumbing is exercised without a GPU.
"""
import torch
import torch.distributed as dist


def shard_region(dims, rank, nranks):
    """Mirror of the C++ rule in smk_set_shard: bit 0 of the rank splits x, bit 1 y, bit 2 z at
    the midpoint (the MetaVolume::brick 2x2x2 grid).  Returns (g0, g1) voxel index bounds."""
    g0, g1 = [0, 0, 0], list(dims)
    bit, n = 0, nranks
    while n > 1:
        a, half = bit % 3, dims[bit % 3] // 2
        if (rank >> bit) & 1:
            g0[a] = max(g0[a], half)
        else:
            g1[a] = min(g1[a], half)
        n >>= 1
        bit += 1
    return tuple(g0), tuple(g1)


def front_to_back_order(eye_voxel, dims, nranks):
    """BSP visibility order of the shards for an eye at `eye_voxel` (voxel index space): per
    split axis the half containing the eye comes first.  Agrees with the reference's
    centre-distance sort for equal bricks (NV20VolRen3D.cpp:195-212) and is exact for every ray."""
    near = [1 if eye_voxel[a] >= dims[a] // 2 - 0.5 else 0 for a in range(3)]
    nbits = nranks.bit_length() - 1
    keyed = []
    for r in range(nranks):
        key = 0
        for bit in range(nbits):
            if ((r >> bit) & 1) != near[bit % 3]:
                key |= 1 << bit
        keyed.append((key, r))
    return [r for _, r in sorted(keyed)]


def tile_pixels(npix, nranks):
    """pixels per rank tile (the frame is padded up to a multiple of nranks)"""
    return (npix + nranks - 1) // nranks


def exchange_and_composite(partial, order, compositor, group=None, recv=None, via_host=False):
    """partial: [npix_padded, 4] premultiplied RGBA of THIS rank's shard (npix_padded divisible
    by the world size).  Returns this rank's finished tile [npix_padded/P, 4].  `recv` lets a
    caller keep one receive buffer per frame slot; `via_host` stages device tensors through host
    memory for a backend that only moves CPU tensors (gloo rehearsals of the GPU plumbing)."""
    P = dist.get_world_size(group)
    n = partial.shape[0]
    assert n % P == 0
    # direct send: piece r of my partial image goes to rank r; I receive piece `me` of everyone
    if via_host and partial.is_cuda:
        send = partial.cpu()
        got = torch.empty_like(send)
        dist.all_to_all_single(got, send, group=group)
        recv = got.to(partial.device)
    else:
        if recv is None:
            recv = torch.empty_like(partial)
        dist.all_to_all_single(recv, partial, group=group)
    layers = recv.view(P, n // P, 4)
    return compositor(layers, order)


def gather_frame(tile, dst=0, group=None, via_host=False, into=None):
    """finished tiles -> full frame on rank dst ([npix_padded,4]); None elsewhere.  `into` (rank dst):
    a preallocated [P, tile, 4] buffer the tiles are received into (no allocation per frame)."""
    P = dist.get_world_size(group)
    me = dist.get_rank(group)
    src = tile.cpu() if via_host and tile.is_cuda else tile
    out = None
    if me == dst:
        if into is not None and not (via_host and tile.is_cuda):
            out = list(into.unbind(0))
        else:
            out = [torch.empty_like(src) for _ in range(P)]
    dist.gather(src, out, dst=dst, group=group)
    if me != dst:
        return None
    if into is not None and not (via_host and tile.is_cuda):
        return into.view(-1, 4)
    return torch.cat(out, 0).to(tile.device)


class Pipeline:
    """Frames back to back with two in flight: while frame i's layers cross xGMI and are merged,
    frame i+1 is already ray-marching.  Each slot has its own stream and buffers; the renderer
    context itself is used by one frame at a time (frame i+1's launch waits for frame i's
    ray-marcher, not for its exchange).  `render(ptr, stream_handle)` ray-marches this rank's
    shard into the [npix_padded,4] buffer at `ptr`; `compositor(layers, order, out, stream_handle)`
    merges [P, tile, 4] front to back into `out`."""

    def __init__(self, render, compositor, npix, group=None, slots=2, via_host=False, frame_check=None):
        """frame_check(token, ptr, stream_handle) -> bool (token = what render() returned), called once a frame's ray-marcher has finished and
        BEFORE its layer is exchanged: True = the frame was invalid and has been rendered again into
        the same buffer (the product passes Renderer.frame_failed + a gather-kernel re-render).  The
        repair is local to the rank, so no rank ever leaves the others waiting in a collective."""
        self.render, self.compositor, self.group, self.via_host = render, compositor, group, via_host
        self.frame_check = frame_check
        P = dist.get_world_size(group)
        me = dist.get_rank(group)
        tp = tile_pixels(npix, P)
        self.npix = npix
        self.slots = []
        for _ in range(slots):
            self.slots.append({
                "stream": torch.cuda.Stream(),
                "partial": torch.zeros((tp * P, 4), dtype=torch.float32, device="cuda"),
                "recv": torch.empty((tp * P, 4), dtype=torch.float32, device="cuda"),
                "tile": torch.zeros((tp, 4), dtype=torch.float32, device="cuda"),
                # rank 0: where the finished tiles of a frame are gathered (allocated once)
                "full": torch.empty((P, tp, 4), dtype=torch.float32, device="cuda") if me == 0 else None})
        self.count = 0
        self.repaired = 0        # frames frame_check had to render again
        self.marched = None      # event: the latest frame's ray-marcher has finished
        self.delivered = None    # event: the latest frame has been copied into the caller's buffer
        self.pending = None      # the frame whose layer has not been exchanged yet

    def frame(self, out, order):
        """enqueue one frame; on rank 0 `out` ([npix,4]) receives it.  Returns at once.  The frame's
        layer is exchanged when the NEXT frame has been enqueued (or at drain()): its ray-marcher has
        then had a frame's time to run, so looking at its status costs the host no wait to speak of,
        and the exchange still overlaps the next frame's ray-marching."""
        sl = self.slots[self.count % len(self.slots)]
        self.count += 1
        s = sl["stream"]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if self.marched is not None:
                s.wait_event(self.marched)
            sl["token"] = self.render(sl["partial"].data_ptr(), s.cuda_stream)   # (whatever identifies the frame to frame_check)
            self.marched = torch.cuda.Event()
            self.marched.record(s)
        sl["marched"], sl["out"], sl["order"] = self.marched, out, order
        prev, self.pending = self.pending, sl
        if prev is not None:
            self._exchange(prev)

    def _exchange(self, sl):
        s = sl["stream"]
        with torch.cuda.stream(s):
            if self.frame_check is not None:
                sl["marched"].synchronize()
                if self.frame_check(sl["token"], sl["partial"].data_ptr(), s.cuda_stream):
                    self.repaired += 1

            def comp(layers, order_):
                self.compositor(layers, order_, sl["tile"], s.cuda_stream)
                return sl["tile"]
            tile = exchange_and_composite(sl["partial"], sl["order"], comp, self.group, sl["recv"], self.via_host)
            full = gather_frame(tile, 0, self.group, self.via_host, into=sl["full"])
            if full is not None:
                if self.delivered is not None:
                    s.wait_event(self.delivered)
                sl["out"].copy_(full[:self.npix])
                self.delivered = torch.cuda.Event()
                self.delivered.record(s)

    def drain(self):
        """exchange the last frame and make the caller's current stream wait for every frame enqueued so far"""
        if self.pending is not None:
            self._exchange(self.pending)
            self.pending = None
        for sl in self.slots:
            torch.cuda.current_stream().wait_stream(sl["stream"])


class ExchangePipeline:
    """Pipeline's contract with the merge behind the C ABI (smk_exchange_*, RCCL transport: grouped
    ncclSend/ncclRecv direct send, ordered over, gather -- csrc/smk_exchange.hip).  Python only
    sequences the calls; no torch collective is in the data path."""

    def __init__(self, render, exchange, npix, rank, frame_check=None):
        self.render, self.x, self.npix, self.rank, self.frame_check = render, exchange, npix, rank, frame_check
        self.count = 0
        self.repaired = 0
        self.pending = None

    def frame(self, out, order=None):
        slot = self.count & 1
        self.count += 1
        st = torch.cuda.current_stream().cuda_stream
        self.x.acquire(slot, st)
        token = self.render(self.x.partial(slot), st)
        self.x.rendered(slot, st)      # (also takes the shards' visibility order under THIS frame's camera)
        if order is not None:
            self.x.set_order(slot, order)
        ev = torch.cuda.Event()
        ev.record()
        prev, self.pending = self.pending, (slot, token, ev, out)
        if prev is not None:
            self._exchange(prev)

    def _exchange(self, p):
        slot, token, ev, out = p
        if self.frame_check is not None:
            ev.synchronize()
            st = torch.cuda.current_stream().cuda_stream
            if self.frame_check(token, self.x.partial(slot), st):
                self.repaired += 1
                self.x.rendered(slot, st)
        self.x.frame(slot, out.data_ptr() if self.rank == 0 else None)

    def drain(self):
        if self.pending is not None:
            self._exchange(self.pending)
            self.pending = None
        self.x.wait(torch.cuda.current_stream().cuda_stream)
