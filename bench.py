#!/usr/bin/env python3
"""bench.py — end-to-end det+rec throughput on MI355X (BASELINE.json metric: pages/sec, A4@200DPI).

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU):
  a "step" = one pass of the hot path over one batch of 64 synthetic A4@200DPI pages (uint8 [64,2339,1654,3],
  already resident in HBM): LANCZOS resize to the reference's 2000-px cap (1414x2000) + contrast/sharpness
  -> DBNet-R18vd (zero-padded to 1440x2016) -> DB post-process -> crops -> CRNN-MV3 + CTC -> decoded strings.
  N GPUs: one process per GPU, 64 pages per rank per step (weak scaling), one RCCL all-gather of the boxes per step.
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
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=PAGES_PER_RANK, help="pages per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--det-sub-batch", type=int, default=16)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lumina_ocr import arch
    from lumina_ocr.dist import all_gather_pages
    from lumina_ocr.engine import Engine
    from lumina_ocr.pipeline import OcrPipeline

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    distributed = world > 1 or "RANK" in os.environ   # under torchrun the RCCL path runs even at world size 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    det_w, rec_w = arch.make_det_weights(1234), arch.make_rec_weights(4321)
    eng = Engine(local_rank)            # raises if liblumina_ocr.so is missing: there is no fallback path
    eng.load_det(det_w)
    eng.load_rec(rec_w)
    eng.set_option("det_sub_batch", args.det_sub_batch)
    pipe = OcrPipeline(eng, post=arch.TEXT_PATH_POST)
    pages = make_pages(torch, args.pages, 2024 + 1000 * rank, device)

    side = torch.cuda.Stream(device) if distributed else None   # result gather runs beside the next step's detection kernels

    def gather(dets):
        return all_gather_pages(dets, pipe.charset, device=device, pages_per_rank=args.pages, stream=side) if distributed else dets

    def run_steps(k):
        """k steps through OcrPipeline.run_many: step i's host-side string decode (and result gather) overlaps the device
        work of step i+1; every step's work, including the last decode, is finished when this returns."""
        dets = None
        for d, _ in pipe.run_many(pages for _ in range(k)):
            dets = gather(d)
        return dets

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
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

    # ---- roofline: HIP events on the launch stream around every conv_mfma launch of one extra step ----
    eng.set_option("time_convs", 1)
    pipe.run(pages)
    rows = eng.conv_timing_detail()          # (layer, kernel instantiation, ms, GFLOP, algorithmic MB) per launch
    eng.set_option("time_convs", 0)
    by_kernel = {}
    for _, kern, ms, gf, mb in rows:
        a = by_kernel.setdefault(kern, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += gf; a[3] += mb
    dom = max(by_kernel, key=lambda k: by_kernel[k][1])
    dn, dms, dgf, dmb = by_kernel[dom]
    fam_ms = sum(a[1] for a in by_kernel.values()); fam_gf = sum(a[2] for a in by_kernel.values())
    achieved = dgf / dms if dms > 0 else 0.0   # GFLOP/ms == TFLOP/s
    traffic = None
    try:  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/r01_pmc_hbm.json, 16-page det forward)
        pmc = json.loads((ROOT / "profiles" / "r01_pmc_hbm.json").read_text())
        want = dom.replace(" ", "")
        for name, v in pmc.get("all", pmc).items():     # rocprofv3 prints "conv_ring_kernel<1, false>(ConvParams, ...)"
            flat = name.replace(" ", "").replace(",false,false>", ">")
            if want in flat:
                traffic = v.get("hbm_bytes_per_launch")
    except Exception:
        pass

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
                       "det_input_px": [hp, wp], "lines_last_step": n_lines, "parallelism": "pages sharded dp%d, 1 all-gather/step" % world},
            "roofline": {"bound": "mfma", "kernel": dom, "launches_per_step": dn, "avg_launch_us": round(dms / dn * 1e3, 1),
                         "flop_per_launch": dgf / dn * 1e9, "algorithmic_bytes_per_launch": dmb / dn * 1e6,
                         "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": "mean HBM bytes/launch of this kernel in a 16-page det forward (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_hbm.json); bench launches cover 16-page sub-batches too",
                         "family": {"kernel": "all conv launches (conv_ring_kernel + conv_mfma_kernel instantiations + conv_pw_kernel)", "launches_per_step": len(rows),
                                    "ms_per_step": round(fam_ms, 3), "achieved": round(fam_gf / fam_ms, 2),
                                    "frac": round(fam_gf / fam_ms / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)}},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU port is timed at N = 1 only (13 s of host work)
            try:
                out["cpu_baseline"] = cpu_baseline(det_w, rec_w, pipe.charset)
            except Exception as e:  # the baseline is informational; never hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "pages/sec", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
