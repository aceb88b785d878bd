"""Oracle (test infrastructure): torch-CPU restatement of the det and rec networks.

No reference symbol exists for these (the engine slot is
/root/reference/backend/services/ocr_service.py:213-246 and
ocr_service_paddleocr_backup.py:285).  "parity unpinned": this file DEFINES the arithmetic.

Arithmetic definition (mode "bf16", what the HIP path implements):
  * every stored activation and every conv/linear weight is bf16 (round-to-nearest-even);
  * contractions accumulate in fp32, bias / residual / activation are applied in fp32,
    the result is rounded to bf16 once;
  * det input: xn = bf16(float(u8) * scale_c + shift_c), scale_c = 1/(255*std_c),
    shift_c = -mean_c/std_c (two fp32 ops, no fma); pixels outside the real page are 0;
  * LSTM: c in fp32, h rounded to bf16 every step; CTC logits fp32, never stored.
Mode "fp32" keeps activations in fp32 (weights stay bf16-exact) and is used to report the
precision cost of bf16 storage, not as the parity target.
"""
from __future__ import annotations

import sys
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "ocr-system_amd"))
from lumina_ocr import arch  # noqa: E402  (layer tables only)


def _rb(x: torch.Tensor, mode: str) -> torch.Tensor:
    """One rounding of a stored tensor: bf16 (the engine's default storage type), f16 (SVTR fp16 mode) or none (mode "fp32")."""
    if mode == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if mode == "f16":
        return x.to(torch.float16).to(torch.float32)
    return x


def _act(x: torch.Tensor, act: str) -> torch.Tensor:
    if act == "relu":
        return torch.relu(x)
    if act == "hswish":
        return x * torch.clamp(x + 3.0, 0.0, 6.0) * (1.0 / 6.0)  # multiply by fp32(1/6): the engine's definition (csrc/common.h apply_act)
    if act == "hsigmoid":  # PaddleOCR SE: slope 0.2, offset 0.5
        return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)
    if act == "sigmoid":
        return torch.sigmoid(x)
    if act == "gelu":  # exact (erf) GELU, fp32
        return F.gelu(x)
    assert act == "none", act
    return x


def _w(wd: Dict[str, np.ndarray], name: str) -> torch.Tensor:
    """OHWI numpy -> OIHW torch."""
    return torch.from_numpy(np.ascontiguousarray(wd[name + ".w"])).permute(0, 3, 1, 2).contiguous()


def _b(wd, name):
    return torch.from_numpy(np.ascontiguousarray(wd[name + ".b"]))


def conv_bn_act(x, wd, name, stride=1, act="none", residual=None, mode="bf16", pad=None, groups=1):
    w = _w(wd, name)
    k = w.shape[-1]
    if pad is None:
        pad = k // 2 if k % 2 == 1 else 0
    if isinstance(stride, int):
        stride = (stride, stride)
    y = F.conv2d(x, w, None, stride=stride, padding=pad, groups=groups)
    y = y + _b(wd, name).view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    return _rb(_act(y, act), mode)


# --------------------------------------------------------------------------------------
# detection
# --------------------------------------------------------------------------------------


def det_normalize(pages_u8: np.ndarray, hp: int, wp: int, mode="bf16") -> torch.Tensor:
    """[B,H,W,3] u8 -> [B,3,hp,wp] normalised, zero outside the real page."""
    b, h, w, _ = pages_u8.shape
    mean = np.asarray(arch.DET_MEAN, np.float32)
    std = np.asarray(arch.DET_STD, np.float32)
    scale = (np.float32(1.0) / (np.float32(255.0) * std)).astype(np.float32)
    shift = (-mean / std).astype(np.float32)
    x = pages_u8.astype(np.float32) * scale  # fp32 mul
    x = x + shift  # fp32 add (separate rounding: no fma)
    out = np.zeros((b, hp, wp, 3), np.float32)
    out[:, :h, :w] = x
    t = torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()
    return _rb(t, mode)


def compose_fpn_p2(wd: Dict[str, np.ndarray]):
    """The lateral fpn.in2 (1x1, no bias in DBFPN) composed into the smoothing conv fpn.p2:
        p2 = conv3x3(W_p2, W_in2 c2 + up2(out3)) = conv3x3(W_c, c2) + conv3x3(W_p2, up2(out3)),  W_c[co,tap,ci] = sum_m W_p2[co,tap,m] W_in2[m,ci]
    -> OHWI weights [64,3,3,320] over the channels [c2 | up2(out3)], or None when fpn.in2 carries a bias (zero padding would then make
    the composition inexact at the border: the engine keeps the two-kernel path, csrc/engine.hip compose_fpn_p2).
    W_c is summed in fp64 in m order (products of bf16 values are exact in fp64), rounded to fp32, then to bf16 — the engine's loader
    does the same operations in the same order."""
    if np.any(wd["fpn.in2.b"] != 0):
        return None
    w_in2 = wd["fpn.in2.w"].reshape(256, 64).astype(np.float64)
    w_p2 = wd["fpn.p2.w"].astype(np.float64)                       # [64,3,3,256]
    acc = np.zeros((64, 3, 3, 64), np.float64)
    for m in range(256):
        acc += w_p2[:, :, :, m, None] * w_in2[m][None, None, None, :]
    wc = arch.bf16_round(acc.astype(np.float32))
    return np.concatenate([wc, wd["fpn.p2.w"].astype(np.float32)], axis=3)


def det_forward(wd: Dict[str, np.ndarray], pages_u8: np.ndarray, hp: Optional[int] = None,
                wp: Optional[int] = None, mode: str = "bf16", taps: Optional[dict] = None, compose: bool = True) -> np.ndarray:
    """pages [B,H,W,3] u8 -> probability map [B,hp,wp] float32 (bf16-exact in mode bf16).
    compose: fpn.p2 through the composed weights (compose_fpn_p2; the engine's default) — the lateral sum out2 is then never
    rounded to bf16; False: the two-step definition (engine option fpn_compose = 0)."""
    b, h, w, _ = pages_u8.shape
    hp = hp or (h + 31) // 32 * 32
    wp = wp or (w + 31) // 32 * 32
    x = det_normalize(pages_u8, hp, wp, mode)

    def tap(name, t):
        if taps is not None:
            taps[name] = t.permute(0, 2, 3, 1).contiguous().numpy()  # NHWC

    with torch.no_grad():
        x = conv_bn_act(x, wd, "stem.conv1", 2, "relu", mode=mode); tap("stem.conv1", x)
        x = conv_bn_act(x, wd, "stem.conv2", 1, "relu", mode=mode); tap("stem.conv2", x)
        x = conv_bn_act(x, wd, "stem.conv3", 1, "relu", mode=mode); tap("stem.conv3", x)
        x = F.max_pool2d(x, 3, 2, 1); tap("stem.pool", x)
        feats = []
        for i in range(4):
            for j in range(2):
                p = f"s{i}.b{j}"
                stride = 2 if (i > 0 and j == 0) else 1
                y = conv_bn_act(x, wd, p + ".conv0", stride, "relu", mode=mode)
                if j == 0:
                    if i == 0:
                        sc = conv_bn_act(x, wd, p + ".short", 1, "none", mode=mode)
                    else:  # vd shortcut: avg-pool 2x2/s2 + 1x1, fused as a 2x2/s2 conv (taps = w/4)
                        sc = conv_bn_act(x, wd, p + ".short", 2, "none", mode=mode, pad=0)
                else:
                    sc = x
                x = conv_bn_act(y, wd, p + ".conv1", 1, "relu", residual=sc, mode=mode)
                tap(p, x)
            feats.append(x)
        c2, c3, c4, c5 = feats
        in5 = conv_bn_act(c5, wd, "fpn.in5", 1, "none", mode=mode)
        out4 = conv_bn_act(c4, wd, "fpn.in4", 1, "none", residual=F.interpolate(in5, scale_factor=2, mode="nearest"), mode=mode)
        out3 = conv_bn_act(c3, wd, "fpn.in3", 1, "none", residual=F.interpolate(out4, scale_factor=2, mode="nearest"), mode=mode)
        p5 = conv_bn_act(in5, wd, "fpn.p5", 1, "none", mode=mode); tap("fpn.p5", p5)
        p4 = conv_bn_act(out4, wd, "fpn.p4", 1, "none", mode=mode); tap("fpn.p4", p4)
        p3 = conv_bn_act(out3, wd, "fpn.p3", 1, "none", mode=mode); tap("fpn.p3", p3)
        wc = compose_fpn_p2(wd) if compose else None
        if wc is not None:     # ONE 3x3 conv over [c2 | up2(out3)] (320 channels), fp32 accumulate, one rounding
            cat = torch.cat([c2, F.interpolate(out3, scale_factor=2, mode="nearest")], dim=1)
            p2 = conv_bn_act(cat, {"fpn.p2c.w": wc, "fpn.p2c.b": wd["fpn.p2.b"]}, "fpn.p2c", 1, "none", mode=mode)
        else:
            out2 = conv_bn_act(c2, wd, "fpn.in2", 1, "none", residual=F.interpolate(out3, scale_factor=2, mode="nearest"), mode=mode)
            p2 = conv_bn_act(out2, wd, "fpn.p2", 1, "none", mode=mode)
        tap("fpn.p2", p2)
        fuse = torch.cat([F.interpolate(p5, scale_factor=8, mode="nearest"),
                          F.interpolate(p4, scale_factor=4, mode="nearest"),
                          F.interpolate(p3, scale_factor=2, mode="nearest"), p2], dim=1)
        tap("fpn.fuse", fuse)
        y = conv_bn_act(fuse, wd, "head.conv1", 1, "relu", mode=mode); tap("head.conv1", y)
        y = _convt2x2(y, wd, "head.convt2", "relu", mode); tap("head.convt2", y)
        y = _convt2x2(y, wd, "head.convt3", "sigmoid", mode)
    return y[:, 0].contiguous().numpy()


def _convt2x2(x, wd, name, act, mode):
    """2x2/s2 transposed conv as a 1x1 conv to 4*cout followed by a pixel shuffle.
    weight rows are ordered (dy*2+dx)*cout + co (arch.make_det_weights)."""
    w = _w(wd, name)  # [4*cout, cin, 1, 1]
    cout = w.shape[0] // 4
    y = F.conv2d(x, w)  # [B, 4*cout, H, W]
    bsz, _, h, ww = y.shape
    y = y.view(bsz, 2, 2, cout, h, ww).permute(0, 3, 4, 1, 5, 2).reshape(bsz, cout, 2 * h, 2 * ww)
    y = y + _b(wd, name).view(1, -1, 1, 1)
    return _rb(_act(y, act), mode)


# --------------------------------------------------------------------------------------
# recognition
# --------------------------------------------------------------------------------------


def rec_normalize(crops_u8: np.ndarray, mode="bf16") -> torch.Tensor:
    """[N,32,320,3] u8 -> [N,3,32,320]: (x/255-0.5)/0.5 as x*(2/255) + (-1), two fp32 ops."""
    x = crops_u8.astype(np.float32) * np.float32(2.0 / 255.0)
    x = x + np.float32(-1.0)
    return _rb(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous(), mode)


def rec_backbone(wd, x: torch.Tensor, mode="bf16", taps=None) -> torch.Tensor:
    def tap(name, t):
        if taps is not None:
            taps[name] = t.permute(0, 2, 3, 1).contiguous().numpy()

    x = conv_bn_act(x, wd, "rec.conv1", 2, "hswish", mode=mode); tap("rec.conv1", x)
    for b in arch.rec_block_table():
        p = f"rec.b{b['idx']}"
        y = conv_bn_act(x, wd, p + ".expand", 1, b["act"], mode=mode)
        y = conv_bn_act(y, wd, p + ".dw", (b["stride_h"], 1), b["act"], mode=mode, groups=b["exp"])
        if b["se"]:
            s = _rb(y.mean(dim=(2, 3), keepdim=True), mode)
            s = conv_bn_act(s, wd, p + ".se1", 1, "relu", mode=mode)
            s = conv_bn_act(s, wd, p + ".se2", 1, "hsigmoid", mode=mode)
            y = _rb(y * s, mode)
        x = conv_bn_act(y, wd, p + ".project", 1, "none", residual=x if b["res"] else None, mode=mode)
        tap(p, x)
    x = conv_bn_act(x, wd, "rec.conv2", 1, "hswish", mode=mode); tap("rec.conv2", x)
    x = F.max_pool2d(x, 2, 2)
    return x  # [N, 288, 1, 80]


def lstm_dir(xs: torch.Tensor, w_ih, w_hh, b, reverse: bool, mode="bf16") -> torch.Tensor:
    """xs [N,T,D] -> hs [N,T,H]; gates ordered i,f,g,o. xproj is rounded to bf16 (stored tensor)."""
    n, t, _ = xs.shape
    h = w_hh.shape[1]
    xp = _rb(xs @ w_ih.t() + b, mode)  # [N,T,4H]
    hs = torch.zeros(n, t, h)
    ht = torch.zeros(n, h)
    ct = torch.zeros(n, h)
    order = range(t - 1, -1, -1) if reverse else range(t)
    for step in order:
        pre = xp[:, step] + ht @ w_hh.t()
        i, f, g, o = pre.split(h, dim=1)
        ct = torch.sigmoid(f) * ct + torch.sigmoid(i) * torch.tanh(g)
        ht = _rb(torch.sigmoid(o) * torch.tanh(ct), mode)
        hs[:, step] = ht
    return hs


def rec_head(wd, feat: torch.Tensor, mode="bf16"):
    """feat [N,288,1,80] -> (argmax idx [N,T] int64, max prob [N,T] f32, logits [N,T,C] f32)."""
    seq = feat.squeeze(2).permute(0, 2, 1).contiguous()  # [N,T,288]
    for layer in (0, 1):
        outs = []
        for d, rev in (("fw", False), ("bw", True)):
            p = f"lstm.l{layer}.{d}"
            outs.append(lstm_dir(seq, torch.from_numpy(wd[p + ".w_ih"]), torch.from_numpy(wd[p + ".w_hh"]),
                                 torch.from_numpy(wd[p + ".b"]), rev, mode))
        seq = torch.cat(outs, dim=2)
    logits = seq @ torch.from_numpy(wd["ctc.fc.w"]).t() + torch.from_numpy(wd["ctc.fc.b"])
    mx, idx = logits.max(dim=2)
    prob = 1.0 / torch.exp(logits - mx.unsqueeze(2)).sum(dim=2)
    return idx.numpy(), prob.numpy().astype(np.float32), logits.numpy(), seq.numpy()


def rec_forward(wd, crops_u8: np.ndarray, mode="bf16", taps=None):
    with torch.no_grad():
        x = rec_normalize(crops_u8, mode)
        feat = rec_backbone(wd, x, mode, taps)
        if taps is not None:
            taps["rec.feat"] = feat.squeeze(2).permute(0, 2, 1).contiguous().numpy()
        return rec_head(wd, feat, mode)


# --------------------------------------------------------------------------------------
# recognition, second family: SVTR-Tiny (arch.svtr_block_table; "parity unpinned": the restatement defines the arithmetic)
# Stored tensors are bf16; every linear / conv accumulates in fp32 and rounds once after bias (+ residual) (+ activation);
# attention scores, soft-max and the probability-weighted sum stay fp32 and round once; LayerNorm is fp32 inside, rounds once.
# --------------------------------------------------------------------------------------
def _linear(x, wd, name, act="none", residual=None, mode="bf16"):
    """x [N,T,Cin] -> [N,T,Cout]; weights stored [Cout,1,1,Cin]."""
    w = torch.from_numpy(np.ascontiguousarray(wd[name + ".w"])).reshape(wd[name + ".w"].shape[0], -1)
    y = x @ w.t() + _b(wd, name)
    if residual is not None:
        y = y + residual
    return _rb(_act(y, act), mode)


def _layernorm(x, wd, name, mode="bf16"):
    g, b = torch.from_numpy(wd[name + ".g"]), torch.from_numpy(wd[name + ".b"])
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return _rb((x - mu) / torch.sqrt(var + arch.SVTR_LN_EPS) * g + b, mode)


def _svtr_mask(h, w):
    """[T,T] bool: query (qy,qx) may attend key (ky,kx) iff |ky-qy| <= 3 and |kx-qx| <= 5 (7 x 11 window)."""
    ys, xs = torch.arange(h).repeat_interleave(w), torch.arange(w).repeat(h)
    return ((ys[:, None] - ys[None, :]).abs() <= arch.SVTR_WINDOW[0] // 2) & ((xs[:, None] - xs[None, :]).abs() <= arch.SVTR_WINDOW[1] // 2)


def svtr_backbone(wd, x, mode="bf16", taps=None):
    """x [N,3,32,320] normalised -> sequence [N,80,192].  Variant (Tiny / Base) = wd["svtr.config"] (arch.svtr_config)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t.contiguous().numpy()

    cfg = arch.svtr_config(wd)
    x = conv_bn_act(x, wd, "svtr.pe1", 2, "gelu", mode=mode)
    x = conv_bn_act(x, wd, "svtr.pe2", 2, "gelu", mode=mode)                 # [N,D0,8,80]
    n, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
    t = _rb(t + _rb(torch.from_numpy(wd["svtr.pos.w"]), mode), mode); tap("svtr.embed", t)
    stage = 0
    for b in arch.svtr_block_table(cfg):
        if b["stage"] != stage:                                               # height merging: conv 3x3 stride (2,1) + LayerNorm
            img = t.reshape(n, h, w, c).permute(0, 3, 1, 2)
            img = conv_bn_act(img, wd, f"svtr.sub{stage}", (2, 1), "none", mode=mode)
            n, c, h, w = img.shape
            t = _layernorm(img.permute(0, 2, 3, 1).reshape(n, h * w, c), wd, f"svtr.sub{stage}.ln", mode); tap(f"svtr.sub{stage}", t)
            stage = b["stage"]
        p, heads = f"svtr.b{b['idx']}", b["heads"]
        hd = c // heads
        qkv = _linear(t, wd, p + ".qkv", mode=mode).reshape(n, h * w, 3, heads, hd).permute(2, 0, 3, 1, 4)   # [3,N,heads,T,hd]
        sc = (qkv[0] @ qkv[1].transpose(-1, -2)) * np.float32(hd ** -0.5)
        if b["local"]:
            sc = sc.masked_fill(~_svtr_mask(h, w), float("-inf"))
        att = _rb((torch.softmax(sc, dim=-1) @ qkv[2]).permute(0, 2, 1, 3).reshape(n, h * w, c), mode)
        t = _layernorm(_linear(att, wd, p + ".proj", residual=t, mode=mode), wd, p + ".ln1", mode)
        m = _linear(t, wd, p + ".fc1", act="gelu", mode=mode)
        t = _layernorm(_linear(m, wd, p + ".fc2", residual=t, mode=mode), wd, p + ".ln2", mode); tap(p, t)
    pooled = _rb(t.reshape(n, h, w, c).mean(dim=1), mode)                     # [N,80,C2]: mean over the 2 remaining rows
    seq = _linear(pooled, wd, "svtr.last", act="hswish", mode=mode); tap("svtr.seq", seq)
    return seq


def svtr_forward(wd, crops_u8: np.ndarray, mode=None, taps=None, widths=None):
    """crops [N,32,320,3] u8 -> (argmax idx [N,80], max prob [N,80] f32, logits, seq).  mode None: the storage type the weights'
    svtr.config names ("bf16" / "f16"; weights are bf16-exact values, which fp16 represents exactly)."""
    mode = mode or arch.svtr_config(wd)["dtype"]
    with torch.no_grad():
        x = rec_normalize(crops_u8, mode)
        if widths is not None:
            for i, wv in enumerate(widths):
                x[i, :, :, int(wv):] = 0
        seq = svtr_backbone(wd, x, mode, taps)
        logits = seq @ torch.from_numpy(wd["svtr.ctc.fc.w"]).t() + torch.from_numpy(wd["svtr.ctc.fc.b"])
        mx, idx = logits.max(dim=2)
        prob = 1.0 / torch.exp(logits - mx.unsqueeze(2)).sum(dim=2)
        return idx.numpy(), prob.numpy().astype(np.float32), logits.numpy(), seq.numpy()


def ctc_greedy(idx: np.ndarray, prob: np.ndarray, charset: List[str]):
    """Standard CTC greedy: collapse repeats, drop blank (0); score = mean max-prob of kept steps
    (0.0 for an empty string).  -> list of (text, score)."""
    out = []
    for row_i, row_p in zip(idx, prob):
        chars, ps = [], []
        prev = -1
        for k, p in zip(row_i.tolist(), row_p.tolist()):
            if k != prev and k != 0:
                chars.append(charset[k]); ps.append(p)
            prev = k
        # fp32 sequential sum in time order, one division (the device kernel does the same)
        s = np.float32(0.0)
        for p in ps:
            s = np.float32(s + np.float32(p))
        out.append(("".join(chars), float(np.float32(s / np.float32(len(ps)))) if ps else 0.0))
    return out
