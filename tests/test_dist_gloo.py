"""CPU: the N>1 path with world_size-2 gloo — page sharding + the one all-gather of result buffers."""
import os
import sys
from pathlib import Path

import numpy as np
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _fake_page(i):
    from lumina_ocr.pipeline import PageDetections
    rng = np.random.default_rng(i)
    n = int(rng.integers(0, 6))
    quads = rng.integers(0, 2000, (n, 8)).astype(np.int32)
    texts = ["p%d-l%d %s" % (i, j, "x" * int(rng.integers(0, 9))) for j in range(n)]
    return PageDetections(quads, texts, rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), 1414, 2000)


def _worker(rank, world, port, n_pages, q):
    sys.path.insert(0, str(ROOT / "ocr-system_amd"))
    import torch
    import torch.distributed as dist
    from lumina_ocr import arch
    from lumina_ocr.dist import PageGather, all_gather_pages, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cs = arch.ctc_charset()
    index_of = {ch: i for i, ch in enumerate(cs)}
    a, b = shard_range(n_pages, rank, world)
    ppr = -(-n_pages // world)
    mine = [_fake_page(i) for i in range(a, b)]
    out = all_gather_pages(mine, cs, pages_per_rank=ppr)               # host objects in
    assert int(out.counts.sum()) == sum(len(_fake_page(i).texts) for i in range(n_pages))
    # the same pages through the tensor front end (what the pipeline feeds from device memory)
    counts = np.array([len(p.texts) for p in mine], np.int32)
    n = int(counts.sum())
    text = np.full((n, 80), -1, np.int32); lens = np.zeros(n, np.int32); k = 0
    for p in mine:
        for t in p.texts:
            ids = [index_of[c] for c in t]
            text[k, :len(ids)] = ids; lens[k] = len(ids); k += 1
    cat = lambda xs, shape, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dt)
    g = PageGather(cs, ppr)
    g.begin(counts)
    h = g.submit(counts, torch.from_numpy(cat([p.quads for p in mine], (0, 8), np.int32).reshape(-1, 8)),
                 torch.from_numpy(cat([p.det_scores for p in mine], (0,), np.float32)), torch.from_numpy(text), torch.from_numpy(lens),
                 torch.from_numpy(cat([p.scores for p in mine], (0,), np.float32)))
    out2 = g.finish(h)
    pages1 = [(o["quads"].tolist(), o["texts"], o["scores"].tolist(), o["det_scores"].tolist()) for o in out.pages()]
    pages2 = [(o["quads"].tolist(), o["texts"], o["scores"].tolist(), o["det_scores"].tolist()) for o in out2.pages()]
    assert pages1 == pages2
    try:                                                               # a rank holding more than pages_per_rank pages: a clear error
        all_gather_pages(mine + mine + mine, cs, pages_per_rank=ppr)
        raised = False
    except ValueError as e:
        raised = "pages_per_rank" in str(e)
    q.put((rank, pages1, raised))
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_all_pages_in_order():
    """7 pages over 2 ranks (4 + 3): the short rank's padding page must not show up — page p of the result is global page p."""
    import socket
    world, n_pages = 2, 7
    with socket.socket() as s:                         # a free port, not a fixed one: concurrent runs must not collide
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = {r: pages for r, pages, _ in got}
    assert all(raised for _, _, raised in got)
    for rank in range(world):
        assert len(res[rank]) == n_pages
        for i in range(n_pages):
            ref = _fake_page(i)
            quads, texts, scores, det = res[rank][i]
            assert quads == ref.quads.tolist() and texts == ref.texts and scores == ref.scores.tolist() and det == ref.det_scores.tolist()
    assert res[0] == res[1]
