"""Page-parallel multi-GPU: one process per GPU, pages sharded in contiguous chunks, ONE all-gather of
fixed-capacity result buffers per batch (RCCL over xGMI on GPUs: torch.distributed backend "nccl"; gloo in CPU tests).

The reference has no distributed path: pages are looped serially and their boxes concatenated afterwards
(/root/reference/backend/services/ocr_service.py:620-627, :635-637) — which is exactly what makes pages the shard unit.
Payload is KB-scale: latency-bound, so a single collective per batch (plus one scalar
all-reduce for the capacity) and never one per page.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

REC_T = 80
ROW = 8 + 3 + REC_T  # quad(8) | score bits | det-score bits | text length | class ids (padded -1)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk of [0, n_items) owned by `rank` (first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def encode_text(text: str, index_of: dict) -> List[int]:
    return [index_of[ch] for ch in text][:REC_T]


def pack_pages(pages: Sequence, capacity: int, index_of: dict) -> np.ndarray:
    """pages: PageDetections-like (quads, texts, scores, det_scores[, text_ids, lens]) -> int32 [P, 1 + capacity, ROW];
    row 0 col 0 = count.  With text_ids/lens present (the pipeline's own output) no per-character Python work happens."""
    buf = np.full((len(pages), 1 + capacity, ROW), -1, np.int32)
    for p, pg in enumerate(pages):
        n = min(len(pg.texts), capacity)
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
    """All ranks' results in global page order; strings are decoded lazily (vectorised utf-32), boxes/scores are views."""

    def __init__(self, buf: np.ndarray, charset: Sequence[str]):
        self.buf = buf
        self._cp = np.array([ord(c) for c in charset], dtype="<u4")

    def __len__(self) -> int:
        return self.buf.shape[0]

    @property
    def counts(self) -> np.ndarray:
        return self.buf[:, 0, 0]

    def page(self, p: int) -> dict:
        n = int(self.buf[p, 0, 0])
        rows = self.buf[p, 1:1 + n]
        cps = self._cp[np.maximum(rows[:, 11:], 0)]
        texts = [cps[i, : rows[i, 10]].tobytes().decode("utf-32-le") for i in range(n)]
        return dict(quads=rows[:, :8].copy(), texts=texts, scores=rows[:, 8].copy().view(np.float32),
                    det_scores=rows[:, 9].copy().view(np.float32))

    def pages(self) -> List[dict]:
        return [self.page(p) for p in range(len(self))]


def unpack_pages(buf: np.ndarray, charset: Sequence[str]):
    return GatheredPages(buf, charset).pages()


def all_gather_pages(local_pages: Sequence, charset: Sequence[str], device=None, pages_per_rank: int = 0, stream=None):
    """Gather every rank's per-page results; returns a GatheredPages over ALL pages in global page order.
    Requires torch.distributed to be initialised; every rank must call it with the same pages_per_rank
    (ranks holding fewer pages are padded with empty pages).  `stream`: a side torch.cuda.Stream to run the two small
    collectives and their copies on, so that they neither wait for nor delay compute already queued on the current stream
    (the inputs are host data: there is nothing to wait for)."""
    if stream is not None:
        import torch
        with torch.cuda.stream(stream):
            return all_gather_pages(local_pages, charset, device=device, pages_per_rank=pages_per_rank)
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    index_of = None if all(getattr(p, "text_ids", None) is not None for p in local_pages) else {ch: i for i, ch in enumerate(charset)}
    ppr = pages_per_rank or len(local_pages)
    cap_t = torch.tensor([max([len(p.texts) for p in local_pages] + [1])], dtype=torch.int32, device=device)
    dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)          # scalar: common capacity
    cap = int(cap_t.item())
    local = np.full((ppr, 1 + cap, ROW), -1, np.int32)
    local[:, 0, 0] = 0
    if len(local_pages):
        local[:len(local_pages)] = pack_pages(local_pages, cap, index_of)
    lt = torch.from_numpy(local).to(device) if device is not None else torch.from_numpy(local)
    gathered = torch.empty((world * ppr, 1 + cap, ROW), dtype=torch.int32, device=lt.device)
    dist.all_gather_into_tensor(gathered, lt)             # the one data collective of the batch
    return GatheredPages(gathered.cpu().numpy(), charset)
