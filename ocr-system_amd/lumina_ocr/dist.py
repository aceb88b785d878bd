"""Page-parallel multi-GPU: one process per GPU, pages sharded in contiguous chunks, ONE all-gather of
fixed-capacity result buffers per batch (RCCL over xGMI on GPUs: torch.distributed backend "nccl"; gloo in CPU tests).

The reference has no distributed path: pages are looped serially and their boxes concatenated afterwards
(/root/reference/backend/services/ocr_service.py:620-627, :635-637) — which is exactly what makes pages the shard unit.
Payload is KB-scale: latency-bound, so a single collective per batch (plus one scalar
all-reduce for the capacity) and never one per page.

Two front ends over the same wire format `int32 [pages_per_rank, 1 + cap, ROW]`:
  * `PageGather` (what bench.py and a batch job use): packs the DEVICE tensors the pipeline already holds (class ids, lengths,
    scores, quads) with a handful of index writes — nothing bounces through the host before the collective; the scalar
    all-reduce of the capacity is started as soon as the box counts are known and has long finished when its value is read.
  * `all_gather_pages` (host objects in, e.g. results that already went through the provider): packs with numpy.
A rank with fewer pages than `pages_per_rank` pads with pages whose count is -1; `GatheredPages` drops them, so that
page p of the result IS global page p whatever the shard sizes.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

REC_T = 80
ROW = 8 + 3 + REC_T  # quad(8) | score bits | det-score bits | text length | class ids (padded -1)
PAD_PAGE = -1        # count of a padding page (dropped by GatheredPages)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk of [0, n_items) owned by `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def encode_text(text: str, index_of: dict) -> List[int]:
    return [index_of[ch] for ch in text][:REC_T]


def pack_pages(pages: Sequence, capacity: int, index_of: Optional[dict]) -> np.ndarray:
    """pages: PageDetections-like (quads, texts, scores, det_scores[, text_ids, lens]) -> int32 [P, 1 + capacity, ROW];
    row 0 col 0 = count.  With text_ids/lens present (the pipeline's own output) no per-character Python work happens."""
    buf = np.full((len(pages), 1 + capacity, ROW), -1, np.int32)
    for p, pg in enumerate(pages):
        n = len(pg.texts)
        if n > capacity:
            raise ValueError("page %d holds %d lines, more than the gather capacity %d" % (p, n, capacity))
        buf[p, 0, 0] = n
        if not n:
            continue
        buf[p, 1:1 + n, :8] = np.asarray(pg.quads[:n], np.int32)
        buf[p, 1:1 + n, 8] = np.asarray(pg.scores[:n], np.float32).view(np.int32)
        buf[p, 1:1 + n, 9] = np.asarray(pg.det_scores[:n], np.float32).view(np.int32)
        ids = getattr(pg, "text_ids", None)
        if ids is not None:
            buf[p, 1:1 + n, 10] = np.asarray(pg.lens[:n], np.int32)
            buf[p, 1:1 + n, 11:11 + REC_T] = np.asarray(ids[:n], np.int32)
        else:
            for i in range(n):
                row = encode_text(pg.texts[i], index_of)
                buf[p, 1 + i, 10] = len(row)
                buf[p, 1 + i, 11:11 + len(row)] = row
    return buf


class GatheredPages:
    """All ranks' results in global page order (padding pages of short ranks removed); strings are decoded lazily
    (vectorised utf-32), boxes/scores are views."""

    def __init__(self, buf: np.ndarray, charset: Sequence[str]):
        keep = buf[:, 0, 0] != PAD_PAGE
        self.buf = buf if bool(keep.all()) else buf[keep]
        from .arch import TextDecoder
        self._dec = charset if isinstance(charset, TextDecoder) else TextDecoder(list(charset))

    def __len__(self) -> int:
        return self.buf.shape[0]

    @property
    def counts(self) -> np.ndarray:
        return self.buf[:, 0, 0]

    def page(self, p: int) -> dict:
        n = int(self.buf[p, 0, 0])
        rows = self.buf[p, 1:1 + n]
        texts = self._dec.decode(rows[:, 11:], rows[:, 10])
        return dict(quads=rows[:, :8].copy(), texts=texts, scores=rows[:, 8].copy().view(np.float32),
                    det_scores=rows[:, 9].copy().view(np.float32))

    def pages(self) -> List[dict]:
        return [self.page(p) for p in range(len(self))]


def unpack_pages(buf: np.ndarray, charset: Sequence[str]):
    return GatheredPages(buf, charset).pages()


def _check_ppr(n_local: int, ppr: int) -> None:
    if n_local > ppr:
        raise ValueError("this rank holds %d pages but pages_per_rank is %d: every rank must pass the size of the largest shard"
                         % (n_local, ppr))


class PageGather:
    """Device-side gather of one batch's results.  Usage per batch (all ranks):
        g.begin(counts_host)                       # as soon as the box counts are on the host: starts the capacity all-reduce
        h = g.submit(counts_host, quads, det_scores, text, length, score)   # device tensors [n, ...] in (page, slot) order
        pages = g.finish(h)                        # GatheredPages over all ranks, global page order
    `stream` (GPU only): a side torch.cuda.Stream for the two collectives and the copy to pinned host memory, so that they
    neither wait for nor delay the detection kernels of the next batch already queued on the compute stream."""

    def __init__(self, charset: Sequence[str], pages_per_rank: int, device=None, stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.charset = charset
        self.ppr = int(pages_per_rank)
        self.device = device if device is not None else torch.device("cpu")
        self.stream = stream
        self.world = dist.get_world_size()
        self._cap_t = None

    def _side(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def begin(self, counts_host: np.ndarray) -> None:
        torch = self.torch
        _check_ppr(len(counts_host), self.ppr)
        cap = max(int(counts_host.max()) if len(counts_host) else 0, 1)
        with self._side():
            self._cap_t = torch.tensor([cap], dtype=torch.int32).to(self.device, non_blocking=True)
            self.dist.all_reduce(self._cap_t, op=self.dist.ReduceOp.MAX)      # scalar: common capacity

    def submit(self, counts_host: np.ndarray, quads, det_scores, text, length, score):
        torch = self.torch
        if self._cap_t is None:
            self.begin(counts_host)
        with self._side():                       # read on the SIDE stream: a copy on the compute stream would wait for the recogniser
            cap = int(self._cap_t.item())        # started in begin(): finished while the recogniser was being enqueued
        self._cap_t = None
        b, n = len(counts_host), int(counts_host.sum())
        dev = self.device
        buf = torch.full((self.ppr, 1 + cap, ROW), -1, dtype=torch.int32, device=dev)
        head = np.full(self.ppr, PAD_PAGE, np.int32)
        head[:b] = counts_host
        buf[:, 0, 0] = torch.from_numpy(head).to(dev, non_blocking=True)
        if n:
            page = np.repeat(np.arange(b, dtype=np.int64), counts_host)
            slot = np.arange(n, dtype=np.int64) - np.repeat(np.cumsum(counts_host) - counts_host, counts_host)
            rows = torch.from_numpy(page * (1 + cap) + 1 + slot).to(dev, non_blocking=True)
            flat = buf.view(-1, ROW)
            body = torch.empty((n, ROW), dtype=torch.int32, device=dev)
            body[:, :8] = quads
            body[:, 8] = score.view(torch.int32)
            body[:, 9] = det_scores.view(torch.int32)
            body[:, 10] = length
            body[:, 11:] = text
            flat.index_copy_(0, rows, body)
        gathered = torch.empty((self.world * self.ppr, 1 + cap, ROW), dtype=torch.int32, device=dev)
        event = None
        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            with self._side():
                self.stream.wait_event(ready)
                self.dist.all_gather_into_tensor(gathered, buf)             # the one data collective of the batch
                host = torch.empty(gathered.shape, dtype=torch.int32, pin_memory=True).copy_(gathered, non_blocking=True)
                event = torch.cuda.Event()
                event.record(self.stream)
            buf.record_stream(self.stream); gathered.record_stream(self.stream)
        else:
            self.dist.all_gather_into_tensor(gathered, buf)
            host = gathered.cpu() if gathered.is_cuda else gathered
        return host, event

    def finish(self, handle) -> GatheredPages:
        host, event = handle
        if event is not None:
            event.synchronize()
        elif host.is_cuda:
            host = host.cpu()
        return GatheredPages(host.numpy(), self.charset)


def all_gather_pages(local_pages: Sequence, charset: Sequence[str], device=None, pages_per_rank: int = 0, stream=None):
    """Host objects in: gather every rank's per-page results; returns a GatheredPages over ALL pages in global page order.
    Requires torch.distributed to be initialised; every rank must call it with the same pages_per_rank = size of the largest
    shard (ranks holding fewer pages are padded; the padding is dropped again on arrival)."""
    if stream is not None:
        import torch
        with torch.cuda.stream(stream):
            return all_gather_pages(local_pages, charset, device=device, pages_per_rank=pages_per_rank)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    index_of = None if all(getattr(p, "text_ids", None) is not None for p in local_pages) else {ch: i for i, ch in enumerate(charset)}
    ppr = pages_per_rank or len(local_pages)
    _check_ppr(len(local_pages), ppr)
    cap_t = torch.tensor([max([len(p.texts) for p in local_pages] + [1])], dtype=torch.int32, device=device)
    dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)          # scalar: common capacity
    cap = int(cap_t.item())
    local = np.full((ppr, 1 + cap, ROW), -1, np.int32)
    local[:, 0, 0] = PAD_PAGE
    if len(local_pages):
        local[:len(local_pages)] = pack_pages(local_pages, cap, index_of)
    lt = torch.from_numpy(local).to(device) if device is not None else torch.from_numpy(local)
    gathered = torch.empty((world * ppr, 1 + cap, ROW), dtype=torch.int32, device=lt.device)
    dist.all_gather_into_tensor(gathered, lt)             # the one data collective of the batch
    return GatheredPages(gathered.cpu().numpy(), charset)
