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
    import torch.distributed as dist
    from lumina_ocr import arch
    from lumina_ocr.dist import all_gather_pages, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cs = arch.ctc_charset()
    a, b = shard_range(n_pages, rank, world)
    ppr = -(-n_pages // world)
    out = all_gather_pages([_fake_page(i) for i in range(a, b)], cs, pages_per_rank=ppr)
    assert int(out.counts.sum()) == sum(len(_fake_page(i).texts) for i in range(n_pages))
    q.put((rank, [(o["quads"].tolist(), o["texts"], o["scores"].tolist()) for o in out.pages()]))
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_all_pages_in_order():
    world, n_pages = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29731, n_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, str(ROOT / "ocr-system_amd"))
    from lumina_ocr.dist import shard_range
    ppr = -(-n_pages // world)
    for rank in range(world):
        got = res[rank]
        assert len(got) == world * ppr
        for r in range(world):
            a, b = shard_range(n_pages, r, world)
            for k, i in enumerate(range(a, b)):
                ref = _fake_page(i)
                quads, texts, scores = got[r * ppr + k]
                assert quads == ref.quads.tolist() and texts == ref.texts and scores == ref.scores.tolist()
            for k in range(b - a, ppr):                      # padding pages are empty
                assert got[r * ppr + k][1] == []
    assert res[0] == res[1]
