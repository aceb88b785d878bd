#!/usr/bin/env python3
"""bench.py — end-to-end det+rec throughput on MI355X (BASELINE.json metric: pages/sec, A4@200DPI).

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU):
  a "step" = one pass of the hot path over one batch of 64 synthetic A4@200DPI pages (uint8 [64,2339,1654,3],
  already resident in HBM): LANCZOS resize to the reference's 2000-px cap (1414x2000) + contrast/sharpness
  -> DBNet-R18vd (zero-padded to 1440x2016) -> DB post-process -> crops -> CRNN-MV3 + CTC -> decoded strings.
  N GPUs: one process per GPU, 64 pages per rank per step (weak scaling), one RCCL all-gather of the boxes per step.
  `python bench.py --gpus N` starts the N ranks itself (a launcher process that never touches the GPU); under
  `python -m torch.distributed.run` (RANK set) it is one of the ranks.
Weights are seeded random-init (no checkpoints exist offline); data is synthetic (rendered text + noise).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     the dominant kernel family (conv_mfma implicit-GEMM): algorithmic FLOPs / HIP-event time
  "cpu_baseline": the oracle port of the same path timed on the host cores over a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "ocr-system_amd"))
sys.path.insert(0, str(ROOT))

PAGES_PER_RANK = 64
A4_H, A4_W = 2339, 1654
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters (bf16 and fp16 dense alike)
HBM_PEAK_GBS = 8000.0                  # same guide: HBM3E spec peak
PMC_FILE = "r03_pmc_hbm.json"          # {"source_sha16": {file: sha}, "kernels": {name: {"hbm_bytes_per_launch": ...}}} (tools/pmc_to_json.py)


def make_pages(torch, n, seed, device):
    """n synthetic A4@200DPI pages on the device: 8 rendered text layouts (PIL) + per-page Gaussian noise (sigma 3)."""
    import numpy as np
    from lumina_ocr import synth
    bases = np.stack([synth.synth_page(A4_H, A4_W, seed + k, n_lines=60, noise=0.0)[0] for k in range(8)])
    base_d = torch.from_numpy(bases).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pages = torch.empty((n, A4_H, A4_W, 3), dtype=torch.uint8, device=device)
    for i in range(n):
        noisy = base_d[i % 8].float() + torch.randn((A4_H, A4_W, 3), generator=g, device=device) * 3.0
        pages[i] = noisy.round().clamp_(0, 255).to(torch.uint8)
    return pages


def cpu_baseline(det_w, rec_w, charset, n_pages=6):
    """Oracle port of the same path on the host cores (torch-CPU fp32 convs + C post-process), bounded sample."""
    import numpy as np
    import torch
    from lumina_ocr import synth
    from oracle import pipeline as op
    # the GPU box exposes every host core but grants a share of 16 per GPU: use that share, and say so
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("LUMINA_CPU_BASELINE_THREADS", 16)))
    torch.set_num_threads(cores)
    pages = np.stack([synth.synth_page(A4_H, A4_W, 2024 + k, n_lines=60)[0] for k in range(n_pages)])
    t0 = time.time()
    from lumina_ocr import arch
    out, _ = op.run_pages(det_w, rec_w, pages, charset, mode="fp32", post=arch.TEXT_PATH_POST)
    dt = time.time() - t0
    return {"value": round(n_pages / dt, 4), "unit": "pages/sec", "cores": cores, "kind": "port",
            "sample": "%d A4@200DPI page(s), full path (resize+enhance+det+post+crop+rec+ctc), oracle/ torch-CPU fp32, %d lines found, %.1f s"
                      % (n_pages, sum(len(o["texts"]) for o in out), dt)}


def visible_gpus() -> int:
    """Device count as a CHILD process sees it (the launcher itself never imports torch or touches the GPU)."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with no RANK in the environment: this process becomes the launcher.  It never touches the GPU
    (no torch.cuda call, not even an import of torch): it starts N fresh child processes of this file, one rank per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, polls them and exits non-zero if any of them did — the
    first failing rank ends the others (they would otherwise sit in a collective waiting for it).
    Rank 0 inherits stdout and prints the one JSON line.  (Under `python -m torch.distributed.run` RANK is set and this is skipped.)"""
    import socket
    import subprocess
    if not args.dry_engine:      # one clear message instead of N torch tracebacks on a mis-sized node
        need, have = (1 if args.share_device else args.gpus), visible_gpus()
        if have < need:
            print("bench.py: --gpus %d needs %d visible GPU(s), this node shows %d: nothing was started" % (args.gpus, need, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LUMINA_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, dict(enumerate(procs))
    while live:
        for r, p in list(live.items()):
            c = p.poll()
            if c is None:
                continue
            del live[r]
            if c != 0:
                print("bench.py: rank %d exited with code %d" % (r, c), file=sys.stderr)
                rc = rc or (c if c > 0 else 1)
        if rc and live:          # a rank died: the survivors are blocked in init / all_reduce / barrier — end them (our own children, by PID)
            for p in live.values():
                p.terminate()
            deadline = time.time() + 10.0
            for r, p in live.items():
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
                print("bench.py: rank %d stopped after another rank failed" % r, file=sys.stderr)
            live = {}
        elif live:
            time.sleep(0.05)
    return rc


class DryPipeline:
    """--dry-engine (CPU rehearsal of the N-rank path, tests/test_bench_launcher.py): same submission structure and the same
    PageGather calls as OcrPipeline, with seeded fake recogniser outputs instead of the HIP engine.  Never used for a number."""

    def __init__(self, torch, rank, gather, pages):
        import numpy as np
        self.torch, self.np, self.rank, self.gather, self.pages = torch, np, rank, gather, pages

    def run_many(self, batches):
        np, torch = self.np, self.torch
        for step, _ in enumerate(batches):
            rng = np.random.default_rng(1000 * self.rank + step)
            counts = rng.integers(0, 5, self.pages).astype(np.int32)
            n = int(counts.sum())
            if self.gather is None:      # single process: no collective at all
                from types import SimpleNamespace
                yield [SimpleNamespace(texts=["x"] * int(c)) for c in counts], None
                continue
            self.gather.begin(counts)
            quads = torch.from_numpy(rng.integers(0, 2000, (n, 8)).astype(np.int32))
            text = torch.full((n, 80), -1, dtype=torch.int32)
            text[:, :3] = torch.from_numpy(rng.integers(1, 90, (n, 3)).astype(np.int32))
            length = torch.full((n,), 3, dtype=torch.int32)
            sc = torch.from_numpy(rng.random(n, dtype=np.float32))
            yield self.gather.finish(self.gather.submit(counts, quads, sc, text, length, sc)), None


class H2DFeed:
    """Batches that start in pinned host memory (value_with_h2d): a copy stream fills one of three device buffers per step, the copy
    of step k+1 is issued when step k is handed out, so it runs under step k's kernels; events order copies and consumers."""

    def __init__(self, torch, pages_dev, device):
        self.torch = torch
        self.host = torch.empty(pages_dev.shape, dtype=pages_dev.dtype, pin_memory=True)
        self.host.copy_(pages_dev)
        self.bufs = [torch.empty_like(pages_dev) for _ in range(3)]
        self.copy_stream = torch.cuda.Stream(device)
        self.device = device

    def batches(self, k):
        torch = self.torch
        cur = torch.cuda.current_stream(self.device)
        copied = [None] * 3        # event: the copy into buffer i has landed
        released = []              # released[j]: recorded on the compute stream when batch j+1 was asked for (batch j's reads are enqueued)

        def issue(j):
            b = j % 3
            with torch.cuda.stream(self.copy_stream):
                if j >= 3:                                  # buffer b was read by batch j-3
                    self.copy_stream.wait_event(released[j - 3])
                self.bufs[b].copy_(self.host, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            copied[b] = ev

        issue(0)
        for j in range(k):
            if j + 1 < k:
                issue(j + 1)
            cur.wait_event(copied[j % 3])
            yield self.bufs[j % 3]
            ev = torch.cuda.Event()
            ev.record(cur)                                  # everything that reads batch j is enqueued by now (run_many asks for j+1 after submitting j)
            released.append(ev)


def bench_stage(args):
    """--config 2 / 3 / 5: one stage of the path on its BASELINE configuration, inputs resident in HBM, one JSON line with its own roofline."""
    import torch
    import numpy as np
    from lumina_ocr import arch, synth
    from lumina_ocr.engine import Engine
    rank = int(os.environ.get("RANK", 0))
    if int(os.environ.get("WORLD_SIZE", 1)) != 1 or args.gpus != 1:
        print("bench.py: --config %d is a single-GPU stage measurement" % args.config, file=sys.stderr)
        return 2
    torch.cuda.set_device(0)
    eng = Engine(0)
    if args.config == 2:
        det_w = arch.make_det_weights(1234)
        eng.load_det(det_w)
        n = 32
        eng.set_option("det_sub_batch", n)
        base = np.stack([synth.synth_page(1024, 1024, 1234 + k, n_lines=40)[0] for k in range(8)])
        x = torch.from_numpy(np.concatenate([base] * 4)).cuda()
        out = [None]
        def step():
            out[0] = eng.det_forward(x, out=out[0]) if out[0] is not None else eng.det_forward(x)
        unit, metric, dtype = "pages/sec", "pages/sec DBNet-R18vd detection only, 1024x1024 pages", "bf16"
        workload = "DBNet-R18 detection only, batch=32 1024x1024 synthetic pages, bf16 (BASELINE configs[1]): u8 pages in HBM -> normalise -> backbone + FPN + head -> bf16 probability map"
        bound, data = "mfma", "synthetic (rendered text pages, seeded); weights random-init (seeded) + hand-set text path"
    else:
        n = 512
        rng = np.random.default_rng(4321 if args.config == 3 else 777)
        base = np.stack([synth.synth_crop(rng)[0] for _ in range(64)])
        x = torch.from_numpy(np.concatenate([base] * 8)).cuda()
        if args.config == 3:
            eng.load_rec(arch.make_rec_weights(4321, code_path=True))
            fwd = eng.rec_forward
            metric, dtype = "crops/sec CRNN-MobileNetV3 + BiLSTM + CTC greedy, 32x320 crops", "bf16"
            workload = "CRNN-MobileNetV3 recognition + CTC greedy, batch=512 32x320 line crops (BASELINE configs[2]): u8 crops in HBM -> class ids + lengths + scores"
            bound = "hbm"
        else:
            cs = arch.devanagari_charset()
            eng.load_svtr(arch.make_svtr_weights(num_classes=len(cs), variant="base", dtype="f16"))
            fwd = eng.svtr_forward
            metric, dtype = "crops/sec SVTR-Base (Hindi dictionary) + CTC greedy, 32x320 crops", "f16"
            workload = "SVTR-base multilingual (Devanagari dictionary, %d classes), fp16 MFMA, batch=512 32x320 crops (BASELINE configs[4], one GPU)" % len(cs)
            bound = "hbm"
        def step():
            idx, prob = fwd(x)
            eng.ctc_decode(idx, prob)
        unit, data = "crops/sec", "synthetic (rendered random strings, seeded); weights random-init (seeded), no checkpoints or dictionaries offline"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eng.conv_timing_detail()
    eng.set_option("time_convs", 1)
    step()
    rows = eng.conv_timing_detail()
    eng.set_option("time_convs", 0)
    roof = roofline_from_rows(rows, bound, stage_ms=el / args.steps * 1e3)
    if rank == 0:
        print(json.dumps({"metric": metric, "value": round(n * args.steps / el, 2), "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
                          "data": data, "config": {"workload": workload, "batch": n, "input": "resident in HBM when the timed region starts"},
                          "roofline": roof}), flush=True)
    eng.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=PAGES_PER_RANK, help="pages per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--det-sub-batch", type=int, default=64, help="pages per detector launch group (64: the whole step; +2 %% over 16)")
    ap.add_argument("--deskew", action="store_true", help="run the reference's default-on de-skew step inside the MAIN timed loop too (by default it is measured by a second loop and reported as value_with_deskew)")
    ap.add_argument("--config", type=int, default=4, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[n-1]: 4 = end-to-end det+rec on A4 pages (the metric; default), 2 = DBNet-R18 detection only, batch 32 of "
                         "1024x1024 pages, 3 = CRNN-MV3 + CTC greedy on 512 line crops, 5 = SVTR-Base fp16 + Hindi dictionary on 512 crops")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary timed loops of config 4 (value_with_deskew, value_with_h2d)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU rehearsal, only with --dry-engine")
    ap.add_argument("--dry-engine", action="store_true", help="no GPU, fake recogniser outputs: rehearses launcher + gather on CPU")
    ap.add_argument("--seed-rank", type=int, default=-1, help="(tests) generate the pages of THIS rank (page seed = 2024 + 1000 * rank) whatever RANK says")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(launcher test, --dry-engine only) this rank raises after the process group is up")
    ap.add_argument("--share-device", action="store_true", help="rehearsal on a one-GPU box: every rank runs the real engine on cuda:0 and the "
                    "gather goes through gloo (RCCL refuses two ranks on one device); the line says so, it is not a scaling number")
    args = ap.parse_args()
    if args.share_device:
        args.backend = "gloo"
    if args.backend == "gloo" and not (args.dry_engine or args.share_device):
        ap.error("--backend gloo is the rehearsal of the multi-rank path and needs --dry-engine (CPU) or --share-device (one GPU)")

    # N > 1 without a launcher: become the launcher (before anything touches the GPU)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.config != 4:
        if args.dry_engine or args.share_device:
            ap.error("--config 2 / 3 / 5 are single-stage measurements on the real engine")
        return bench_stage(args)

    import torch
    import torch.distributed as dist
    from lumina_ocr import arch
    from lumina_ocr.dist import PageGather

    rank = int(os.environ.get("RANK", 0))
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: reporting n_gpus=%d (the ranks that actually ran)" % (args.gpus, world, world), file=sys.stderr)
    distributed = world > 1 or "RANK" in os.environ   # under a launcher the collective path runs even at world size 1
    dry = args.dry_engine
    device = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry or args.share_device:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=device)
    ranks_seen = dist.get_world_size() if distributed else 1
    if dry and args.fail_rank == rank:
        raise RuntimeError("rank %d fails on request (--fail-rank)" % rank)

    charset = arch.ctc_charset()
    side = torch.cuda.Stream(device) if (distributed and not dry) else None   # result gather runs beside the next step's detection kernels
    gather = PageGather(charset, args.pages, device=device, stream=side) if distributed else None
    det_w = rec_w = eng = None
    if dry:
        pipe, pages = DryPipeline(torch, rank, gather, args.pages), None
    else:
        from lumina_ocr.engine import Engine
        from lumina_ocr.pipeline import OcrPipeline
        det_w, rec_w = arch.make_det_weights(1234), arch.make_rec_weights(4321, code_path=True)
        eng = Engine(local_rank)            # raises if liblumina_ocr.so is missing: there is no fallback path
        eng.load_det(det_w)
        eng.load_rec(rec_w)
        eng.set_option("det_sub_batch", args.det_sub_batch)
        pipe = OcrPipeline(eng, charset=charset, post=arch.TEXT_PATH_POST, gather=gather)
        pages = make_pages(torch, args.pages, 2024 + 1000 * (args.seed_rank if args.seed_rank >= 0 else rank), device)

    def run_steps(k):
        """k steps through run_many: step i's host-side string decode (multi-GPU: its result gather, on a side stream) overlaps
        the device work of step i+1; every step's work, including the last decode / gather, is finished when this returns."""
        dets = None
        for d, _ in (pipe.run_many((pages for _ in range(k)), deskew=args.deskew) if not dry else pipe.run_many(pages for _ in range(k))):
            dets = d
        return dets

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    dets = run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    dets = run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_lines = int(dets.counts.sum()) if hasattr(dets, "counts") else sum(len(d.texts) for d in dets)
    n_pages_last = len(dets)
    # identity of the gathered pages (small runs only: the multi-rank tests compare it with each rank's pages run alone)
    digests = None
    if n_pages_last <= 16:
        import zlib
        per_page = [pg["texts"] for pg in dets.pages()] if hasattr(dets, "pages") else [list(d.texts) for d in dets]
        digests = ["%08x" % zlib.crc32("\n".join(t).encode("utf-8")) for t in per_page]

    # ---- secondary figures of the same run (the exclusions of the headline, measured instead of described) ----
    secondary = {}
    if not dry and not args.no_secondary:
        def timed(fn):
            """One secondary loop.  Every rank runs the same collectives whatever happens inside fn (fn None / raising = this rank failed):
            a failure yields None on all ranks instead of a rank stuck in a barrier."""
            fence()
            t = time.perf_counter()
            ok = fn is not None
            if ok:
                try:
                    fn()
                except Exception as e:
                    ok = False
                    print("bench.py: secondary loop failed on rank %d: %s" % (rank, e), file=sys.stderr)
            fence()
            el = time.perf_counter() - t if ok else float("inf")
            if distributed:
                tt = torch.tensor([el], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            return round(args.pages * world * args.steps / el, 2) if el != float("inf") else None

        if not args.deskew:   # the reference provider's default: OCR_APPLY_DESKEW = true (backend/config.py:85)
            def with_deskew():
                for _ in pipe.run_many((pages for _ in range(args.steps)), deskew=True):
                    pass
            try:
                for _ in pipe.run_many((pages for _ in range(2)), deskew=True):   # (first use uploads its tables)
                    pass
            except Exception:
                with_deskew = None
            secondary["value_with_deskew"] = timed(with_deskew)
        # pages start in PINNED HOST memory: every step's 64 x 11.6 MB cross PCIe inside the timed region (copy stream, three device
        # buffers: the copy of step k+2 runs under the compute of step k+1)
        feed = None
        def with_h2d():
            for _ in pipe.run_many(feed.batches(args.steps), deskew=args.deskew):
                pass
        try:
            feed = H2DFeed(torch, pages, device)
            for _ in pipe.run_many(feed.batches(2), deskew=args.deskew):
                pass
        except Exception:
            with_h2d = None
        secondary["value_with_h2d"] = timed(with_h2d)
        del feed
        # pages start as JPEG FILES in host memory (written once, outside the timed region, by the engine's own encoder at quality 95:
        # byte-identical to Pillow's files): every step uploads the entropy-coded bytes and decodes them on the device
        dec_status, dec_bufs, sizes_h = [], None, None
        def decoded_batches(k):   # nothing synchronises: the host prepares batch j+1 while the device works on batch j
            for j in range(k):
                out, status = eng.jpeg_decode_async(jfiles, A4_H, A4_W, out=dec_bufs[j % 3])
                dec_status.append(status)
                yield out
        def with_decode():
            for _ in pipe.run_many(decoded_batches(args.steps), deskew=args.deskew):
                pass
            if any(int(st.abs().sum()) for st in dec_status):     # e.g. -5: 12 blind passes were not enough for some page
                raise RuntimeError("device JPEG decode status %s" % [st.tolist() for st in dec_status if int(st.abs().sum())][:1])
        try:
            files_d, sizes_d = eng.jpeg_encode(pages, 95, max_bytes=8 << 20)
            sizes_h = sizes_d.cpu().numpy()
            jfiles = [files_d[i, : int(sizes_h[i])].cpu().numpy().tobytes() for i in range(args.pages)]
            del files_d
            dec_bufs = [torch.empty_like(pages) for _ in range(3)]
            for _ in pipe.run_many(decoded_batches(2), deskew=args.deskew):
                pass
        except Exception as e:
            print("bench.py: JPEG decode loop not set up on rank %d: %s" % (rank, e), file=sys.stderr)
            with_decode = None
        secondary["value_with_decode"] = timed(with_decode)
        if secondary["value_with_decode"] is not None:
            secondary["decode_note"] = "%d JPEG files (quality 95, 4:2:0, %.2f MB mean) per step; %d synchronisation passes per decode" % (
                args.pages, float(sizes_h.mean()) / 1e6, eng.jpeg_last_passes)
        else:
            secondary["decode_note"] = "not measured (see stderr)"
        del dec_bufs

    roofline = None
    if not dry:
        roofline = measure_roofline(eng, pipe, pages)

    if rank == 0:
        hp, wp = 2016, 1440
        total_pages = args.pages * world * args.steps
        out = {
            "metric": "pages/sec end-to-end det+rec, A4@200DPI",
            "value": round(total_pages / elapsed, 2),
            "unit": "pages/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic (rendered text pages + noise, seeded); weights random-init (seeded), no checkpoints offline",
            "config": {"workload": "end-to-end det+rec, batch=64 A4@200DPI pages per GPU (BASELINE configs[3])",
                       "pages_per_gpu": args.pages, "global_batch": args.pages * world, "page_px": [A4_H, A4_W],
                       "det_input_px": [hp, wp], "lines_last_step": n_lines, "pages_gathered_last_step": n_pages_last, "page_digests_last_step": digests,
                       "parallelism": "pages sharded dp%d, 1 all-gather/step" % world + (" (REHEARSAL: all ranks share cuda:0, gather over gloo — not a scaling number)" if args.share_device else ""),
                       "ranks": ranks_seen, "collective_backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if distributed else None,
                       "input": "pages pre-decoded (uint8 RGB) and resident in HBM when the timed region starts",
                       "outside_timed_region": "image decode, host->device copy, JPEG hand-off of the processed page (2.5 ms per 64 pages on the device)",
                       "det_sub_batch": args.det_sub_batch,
                       "deskew": ("on" if args.deskew else "off in `value` (det+rec as BASELINE.json names it; the reference's own deskew is a no-op without OpenCV, "
                                  "image_preprocessing.py:383-385); `value_with_deskew` = the same steps with the provider's default-on de-skew (backend/config.py:85) on the device")},
            "roofline": roofline,
        }
        if secondary:
            out.update(secondary)
            out["secondary_note"] = ("same process, same pages, %d steps each after the main loop: value_with_deskew = + the reference's default-on de-skew step; "
                                     "value_with_h2d = pages start in pinned host memory and are copied to the device inside the timed region; "
                                     "value_with_decode = pages start as JPEG files in host memory, uploaded and decoded on the device inside the timed region" % args.steps)
        if dry:
            out["data"] = "DRY ENGINE (CPU rehearsal of the launcher and the gather; not a measurement)"
            out["value"] = None
        if not args.no_cpu_baseline and world == 1 and not dry:   # the CPU port is timed at N = 1 only (13 s of host work)
            try:
                out["cpu_baseline"] = cpu_baseline(det_w, rec_w, charset)
            except Exception as e:  # the baseline is informational; never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "pages/sec", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def flat_kernel_name(name):
    """the profiler's spelling of a kernel name -> the engine's (no blanks)"""
    return name.replace(" ", "")


def _sha16(path):
    import hashlib
    return hashlib.sha256(Path(path).read_bytes()).hexdigest()[:16]


def roofline_from_rows(rows, bound, stage_ms=None):
    """rows: (layer, kernel instantiation, ms, GFLOP, algorithmic MB) per timed launch of one step -> the roofline object of the kernel
    with the largest summed time.  bound "mfma": algorithmic FLOP / time against the dense MFMA peak; "hbm": algorithmic bytes / time
    against the HBM peak (the recogniser paths: short-K products and depthwise convolutions, < 100 FLOP per byte)."""
    by_kernel = {}
    for _, kern, ms, gf, mb in rows:
        a = by_kernel.setdefault(kern, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += gf; a[3] += mb
    dom = max(by_kernel, key=lambda k: by_kernel[k][1])
    dn, dms, dgf, dmb = by_kernel[dom]
    tot_ms = sum(a[1] for a in by_kernel.values())
    out = {"bound": bound, "kernel": dom, "launches_per_step": dn, "avg_launch_us": round(dms / dn * 1e3, 1), "share_of_timed_launches": round(dms / tot_ms, 3),
           "flop_per_launch": dgf / dn * 1e9, "algorithmic_bytes_per_launch": dmb / dn * 1e6}
    if bound == "mfma":
        ach = dgf / dms
        out.update(achieved=round(ach, 2), peak=MFMA_BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4))
    else:
        ach = dmb / dms   # MB / ms == GB/s
        out.update(achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                   mfma_tflops=round(dgf / dms, 2))
        if dom.startswith("mbconv_kernel"):
            out["limiter"] = "vector ALU: the depthwise K x K convolution is fp32 FMA work (25 taps per output for K = 5), not bytes and not MFMA (PMC, profiles/r01_c_pmc_sq_rec_summary.txt)"
    out["traffic"] = None
    out["all_timed_launches"] = {"launches_per_step": len(rows), "ms_per_step": round(tot_ms, 3), "tflops": round(sum(a[2] for a in by_kernel.values()) / tot_ms, 2),
                                 "algorithmic_gb_per_s": round(sum(a[3] for a in by_kernel.values()) / tot_ms, 1)}
    if stage_ms is not None:
        out["all_timed_launches"]["stage_ms_per_step"] = round(stage_ms, 3)
    return out


def measure_roofline(eng, pipe, pages):
    """HIP events on the launch stream around every conv launch of one extra step (outside the timed region)."""
    eng.set_option("time_convs", 1)
    pipe.finish(pipe.submit_recognize(*pipe.submit_detect(pages)))
    rows = eng.conv_timing_detail()          # (layer, kernel instantiation, ms, GFLOP, algorithmic MB) per launch
    eng.set_option("time_convs", 0)
    rows = [r for r in rows if r[1].startswith("conv_")]   # the conv family (the recogniser's other launches are timed too: --config 3)
    by_kernel = {}
    for _, kern, ms, gf, mb in rows:
        a = by_kernel.setdefault(kern, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += gf; a[3] += mb
    dom = max(by_kernel, key=lambda k: by_kernel[k][1])
    dn, dms, dgf, dmb = by_kernel[dom]
    fam_ms = sum(a[1] for a in by_kernel.values()); fam_gf = sum(a[2] for a in by_kernel.values())
    achieved = dgf / dms if dms > 0 else 0.0   # GFLOP/ms == TFLOP/s
    # HBM bytes per launch: separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950 correction), committed
    # under profiles/ together with the hash of the kernel sources they were taken on; a stale file yields null, never an old number
    traffic, note = None, "no PMC file for the current kernel sources"
    try:
        pmc = json.loads((ROOT / "profiles" / PMC_FILE).read_text())
        srcs = pmc.get("source_sha16", {})
        cur = {k: _sha16(ROOT / "ocr-system_amd" / "csrc" / k) for k in srcs}
        if srcs and cur == srcs:
            want = dom.replace(" ", "")
            for name, v in pmc["kernels"].items():
                if want in flat_kernel_name(name):
                    traffic = v.get("hbm_bytes_per_launch")
                    note = "mean HBM bytes/launch of this kernel over two %d-page det forwards, the bench's launch size (tools/pmc_target.py; profiles/%s): 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes" % (PAGES_PER_RANK, PMC_FILE)
        else:
            note = "profiles/%s was collected on other kernel sources than the ones running: not reported" % PMC_FILE
    except Exception:
        pass
    return {"bound": "mfma", "kernel": dom, "launches_per_step": dn, "avg_launch_us": round(dms / dn * 1e3, 1),
            "flop_per_launch": dgf / dn * 1e9, "algorithmic_bytes_per_launch": dmb / dn * 1e6,
            "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
            "traffic_over_algorithmic": (round(traffic / (dmb / dn * 1e6), 3) if traffic else None), "traffic_note": note,
            "family": {"kernel": "all conv launches (conv_ring_kernel + conv_mfma_kernel instantiations + conv_pw_kernel)", "launches_per_step": len(rows),
                       "ms_per_step": round(fam_ms, 3), "achieved": round(fam_gf / fam_ms, 2),
                       "frac": round(fam_gf / fam_ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}}


if __name__ == "__main__":
    main()
