"""Architecture tables, seeded weights and the weight-blob format of the det+rec engine.

The reference ships no det+rec network: the engine slot is a remote
call (/root/reference/backend/services/ocr_service.py:213-246) or a third-party VLM
(/root/reference/backend/services/ocr_service_paddleocr_backup.py:285).  The networks
here are the ones BASELINE.json names, laid out after the public PaddleOCR model
definitions (det_r18_vd_db; rec_mv3_none_bilstm_ctc): *defined by this build*.

Everything in this file is host-side plumbing shared by the ctypes host, the C++
engine (which hard-codes the same layer names) and the test oracle:
  * DET_LAYERS / REC_* tables  — layer names, shapes, strides, activations
  * make_det_weights / make_rec_weights — deterministic seeded weights, BN folded
  * write_blob / read_blob — the "LOCW" weight container the C-ABI loads
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# bf16 helpers (numpy has no bf16: values are kept as float32 that are exactly bf16)
# --------------------------------------------------------------------------------------


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 -> nearest-even bf16, returned as float32 (exactly representable)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).reshape(x.shape)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32).reshape(b.shape)


# --------------------------------------------------------------------------------------
# Detection: DBNet = ResNet18_vd + DBFPN(256) + DBHead (probability branch)
# --------------------------------------------------------------------------------------
# conv entry: (name, cin, cout, k, stride, act, residual_from, kind)
#   kind: "conv" | "short_vd" (2x2 avg-pool fused as a 2x2/s2 conv) | "convt" (2x2/s2 transposed)
#   act : "relu" | "none" | "sigmoid"
DET_MEAN = (0.485, 0.456, 0.406)
DET_STD = (0.229, 0.224, 0.225)
DET_STAGE_CH = (64, 128, 256, 512)
DET_FPN_CH = 256
DET_THRESH = 0.3
DET_BOX_THRESH = 0.6
DET_UNCLIP_RATIO = 1.5
DET_MIN_SIZE = 3
DET_MAX_CANDIDATES = 1000


def det_conv_table() -> List[dict]:
    t: List[dict] = []

    def add(name, cin, cout, k, stride, act, kind="conv"):
        t.append(dict(name=name, cin=cin, cout=cout, k=k, stride=stride, act=act, kind=kind))

    add("stem.conv1", 3, 32, 3, 2, "relu")
    add("stem.conv2", 32, 32, 3, 1, "relu")
    add("stem.conv3", 32, 64, 3, 1, "relu")
    cin = 64
    for i, ch in enumerate(DET_STAGE_CH):
        for j in range(2):
            stride = 2 if (i > 0 and j == 0) else 1
            add(f"s{i}.b{j}.conv0", cin, ch, 3, stride, "relu")
            add(f"s{i}.b{j}.conv1", ch, ch, 3, 1, "relu")  # relu applied after the residual add
            if j == 0:
                if i == 0:
                    add(f"s{i}.b{j}.short", cin, ch, 1, 1, "none")
                else:
                    add(f"s{i}.b{j}.short", cin, ch, 2, 2, "none", kind="short_vd")
            cin = ch
    for lvl, ch in zip((5, 4, 3, 2), reversed(DET_STAGE_CH)):
        add(f"fpn.in{lvl}", ch, DET_FPN_CH, 1, 1, "none")
    for lvl in (5, 4, 3, 2):
        add(f"fpn.p{lvl}", DET_FPN_CH, DET_FPN_CH // 4, 3, 1, "none")
    add("head.conv1", DET_FPN_CH, 64, 3, 1, "relu")
    add("head.convt2", 64, 64, 2, 2, "relu", kind="convt")
    add("head.convt3", 64, 1, 2, 2, "sigmoid", kind="convt")
    return t


def det_macs_per_page(h: int, w: int) -> int:
    """Algorithmic MACs of one det forward on an h x w page (h, w multiples of 32)."""
    assert h % 32 == 0 and w % 32 == 0
    macs = 0
    res = {"stem.conv1": 2, "stem.conv2": 2, "stem.conv3": 2}
    for e in det_conv_table():
        n = e["name"]
        if n in res:
            s = res[n]
        elif n.startswith("s"):
            s = 4 << int(n[1])
        elif n.startswith("fpn.in") or n.startswith("fpn.p"):
            s = 1 << int(n[-1])
        elif n == "head.conv1":
            s = 4
        elif n == "head.convt2":
            s = 4  # input resolution; 4 output taps each 1x1
        elif n == "head.convt3":
            s = 2
        ho, wo = h // s, w // s
        taps = e["k"] * e["k"]
        if e["kind"] == "convt":
            macs += ho * wo * 4 * e["cin"] * e["cout"]
        elif e["kind"] == "short_vd":
            macs += ho * wo * e["cin"] * e["cout"]  # avg-pool + 1x1 (the algorithmic count)
        else:
            macs += ho * wo * taps * e["cin"] * e["cout"]
    return macs


# --------------------------------------------------------------------------------------
# Recognition: CRNN = MobileNetV3-small x0.5 -> 2 x BiLSTM(96) -> FC -> CTC greedy
# --------------------------------------------------------------------------------------
REC_H, REC_W, REC_T = 32, 320, 80
REC_HIDDEN = 96
REC_FEAT = 288


def _make_div(v: float, d: int = 8) -> int:
    nv = max(d, int(v + d / 2) // d * d)
    if nv < 0.9 * v:
        nv += d
    return nv


# (k, exp, c, se, act, stride_h) at scale 1.0; width stride is always 1
_MV3_SMALL = [
    (3, 16, 16, True, "relu", 1),
    (3, 72, 24, False, "relu", 2),
    (3, 88, 24, False, "relu", 1),
    (5, 96, 40, True, "hswish", 2),
    (5, 240, 40, True, "hswish", 1),
    (5, 240, 40, True, "hswish", 1),
    (5, 120, 48, True, "hswish", 1),
    (5, 144, 48, True, "hswish", 1),
    (5, 288, 96, True, "hswish", 2),
    (5, 576, 96, True, "hswish", 1),
    (5, 576, 96, True, "hswish", 1),
]


def rec_block_table(scale: float = 0.5) -> List[dict]:
    blocks = []
    cin = _make_div(16 * scale)
    for i, (k, exp, c, se, act, sh) in enumerate(_MV3_SMALL):
        e, co = _make_div(exp * scale), _make_div(c * scale)
        blocks.append(dict(idx=i, k=k, cin=cin, exp=e, cout=co, se=se, act=act, stride_h=sh,
                           res=(sh == 1 and cin == co), se_mid=e // 4))
        cin = co
    return blocks


def rec_stem_ch(scale: float = 0.5) -> int:
    return _make_div(16 * scale)


# --------------------------------------------------------------------------------------
# CTC dictionary (the PP-OCR key file is absent offline: the build ships its own list)
# --------------------------------------------------------------------------------------


def ctc_charset(num_classes: int = 6625) -> List[str]:
    """index 0 = blank, last index = space (PP-OCR `use_space_char` convention)."""
    n_chars = num_classes - 2
    chars = [chr(c) for c in range(0x21, 0x7F)]
    cp = 0x4E00
    while len(chars) < n_chars:
        chars.append(chr(cp))
        cp += 1
    return ["\x00"] + chars[:n_chars] + [" "]


class TextDecoder:
    """class ids [n, T] (-1 padded) + lengths [n] -> strings.  Dictionaries whose symbols are single code points (every PP-OCR
    key file) decode vectorised through utf-32; multi-code-point symbols fall back to a join per line."""

    def __init__(self, charset: List[str]):
        self.charset = list(charset)
        self.single = all(len(c) == 1 for c in self.charset)
        self._cp = np.array([ord(c) for c in self.charset], dtype="<u4") if self.single else None

    def decode(self, ids: np.ndarray, lens: np.ndarray) -> List[str]:
        if len(ids) == 0:
            return []
        if self.single:
            cps = self._cp[np.maximum(ids, 0)]
            return [cps[i, : lens[i]].tobytes().decode("utf-32-le") for i in range(len(ids))]
        return ["".join(self.charset[k] for k in ids[i, : lens[i]]) for i in range(len(ids))]


def load_charset(path, use_space_char: bool = True) -> List[str]:
    """Dictionary FILE -> class list.  Format = PP-OCR key files (ppocr_keys_v1.txt, devanagari_dict.txt ...): UTF-8, one symbol
    per line, no header.  Class 0 is the CTC blank, the file's symbols follow in file order, a space is appended last
    (PP-OCR `use_space_char`).  The provider checks len(result) against the recogniser head's class count."""
    from pathlib import Path
    lines = Path(path).read_text(encoding="utf-8").split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    syms = [ln.rstrip("\r") for ln in lines]
    if any(len(s) == 0 for s in syms):
        raise ValueError("dictionary %s holds an empty line" % path)
    if len(set(syms)) != len(syms):
        raise ValueError("dictionary %s holds duplicate symbols" % path)
    return ["\x00"] + syms + ([" "] if use_space_char and " " not in syms else [])


def save_charset(path, charset: List[str]) -> None:
    """Inverse of load_charset for a class list that starts with the blank and ends with the space."""
    from pathlib import Path
    assert charset[0] == "\x00" and charset[-1] == " "
    Path(path).write_text("".join(c + "\n" for c in charset[1:-1]), encoding="utf-8")


def devanagari_charset() -> List[str]:
    """blank + Devanagari block + digits/latin punctuation + space (config 5; build's own list)."""
    chars = [chr(c) for c in range(0x0900, 0x0980)] + [chr(c) for c in range(0x21, 0x7F) if not chr(c).isalpha()]
    return ["\x00"] + chars + [" "]


# --------------------------------------------------------------------------------------
# Seeded weights (BN folded into conv weight + bias; conv weights are bf16-exact float32)
# --------------------------------------------------------------------------------------


def _he(rng, cout, k, cin, gain=1.0, depthwise=False):
    fan_in = k * k * (1 if depthwise else cin)
    w = rng.standard_normal((cout, k, k, cin), dtype=np.float32) * np.float32(gain * np.sqrt(2.0 / fan_in))
    return bf16_round(w)


# Hand-set "text kernel" path (channel 0 of every tensor it crosses), installed on top of the seeded random
# weights so that synthetic pages produce real line boxes (no trained checkpoint exists offline):
#   stem.conv1 ch k<9 : relu(-mean_rgb(xn at tap k) - 0.5)            per-pixel "ink" (darker than mid-gray)
#   stem.conv2 ch 0   : sum of the 9 ink taps                          ink in a 3x3 window
#   stem.conv3 ch 0   : 1x3 horizontal sum / 4 ; max-pool              (1/4 resolution)
#   s0.*              : carried unchanged (shortcut = identity on ch 0, residual branch row 0 = 0)
#   fpn.in2/p2, head  : ch 0 -> 1x3 horizontal smoothing twice -> logit = gain * S - offset
# Rows feeding channel 0 read nothing else; every other row stays random (dense, realistic operands).
TEXT_GAIN, TEXT_OFFSET = 8.0, 26.0
# Post-process parameters that go with the hand-set kernel: it spans the full glyph height and is near-binary, whereas a
# trained DB kernel is shrunk (ratio 0.4) and soft, for which PaddleOCR's defaults (unclip 1.5, box_thresh 0.6) are meant.
TEXT_PATH_POST = dict(unclip_ratio=0.35, box_thresh=0.5)
DEFAULT_POST = dict(unclip_ratio=DET_UNCLIP_RATIO, box_thresh=DET_BOX_THRESH)


def _install_text_path(w: Dict[str, np.ndarray]) -> None:
    def zero_row(name, row=0):
        w[name + ".w"][row] = 0.0
        w[name + ".b"][row % len(w[name + ".b"])] = 0.0

    for k in range(9):
        zero_row("stem.conv1", k)
        w["stem.conv1.w"][k, k // 3, k % 3, :] = -1.0 / 3.0
        w["stem.conv1.b"][k] = -0.5
    zero_row("stem.conv2")
    w["stem.conv2.w"][0, 1, 1, :9] = 1.0
    zero_row("stem.conv3")
    w["stem.conv3.w"][0, 1, :, 0] = 0.25
    # 1/4 resolution: four horizontal 1x3 low-pass stages inside stage 0 (long horizontal runs = text lines survive,
    # short blobs such as descenders bridging two lines are attenuated)
    zero_row("s0.b0.short")
    zero_row("s0.b0.conv0", 1)
    w["s0.b0.conv0.w"][1, 1, :, 0] = 1.0 / 3.0
    zero_row("s0.b0.conv1")
    w["s0.b0.conv1.w"][0, 1, :, 1] = 1.0 / 3.0
    zero_row("s0.b1.conv0", 1)
    w["s0.b1.conv0.w"][1, 1, :, 0] = 1.0 / 3.0
    zero_row("s0.b1.conv1")
    w["s0.b1.conv1.w"][0, 1, :, 1] = 1.0 / 3.0
    for lvl in (3, 4, 5):
        zero_row(f"fpn.in{lvl}")
    zero_row("fpn.in2")
    w["fpn.in2.w"][0, 0, 0, 0] = 1.0
    zero_row("fpn.p2")
    w["fpn.p2.w"][0, 1, :, 0] = 1.0 / 3.0
    zero_row("head.conv1")
    w["head.conv1.w"][0, 1, :, 192] = 1.0 / 3.0
    for q in range(4):
        zero_row("head.convt2", q * 64)
        w["head.convt2.w"][q * 64, 0, 0, 0] = 1.0
        w["head.convt3.w"][q] = 0.0
        w["head.convt3.w"][q, 0, 0, 0] = TEXT_GAIN
    w["head.convt3.b"][:] = -TEXT_OFFSET
    for k in list(w):
        if k.endswith(".w"):
            w[k] = bf16_round(w[k])


def make_det_weights(seed: int = 1234, text_path: bool = True) -> Dict[str, np.ndarray]:
    """name.w: OHWI float32 (bf16-exact), name.b: float32 [cout]. convT: [(dy*2+dx)*cout+co][cin]."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for e in det_conv_table():
        n, cin, cout, k = e["name"], e["cin"], e["cout"], e["k"]
        gain = 1.0
        if n.endswith(".conv1") and n.startswith("s"):
            gain = 0.5  # second conv of a residual branch
        if n.endswith(".short"):
            gain = 0.7071
        if n.startswith("fpn."):
            gain = 0.7071
        if e["kind"] == "convt":
            fan_in = cin
            wt = rng.standard_normal((4 * cout, 1, 1, cin), dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            w[n + ".w"] = bf16_round(wt)
        elif e["kind"] == "short_vd":
            base = rng.standard_normal((cout, 1, 1, cin), dtype=np.float32) * np.float32(gain * np.sqrt(2.0 / cin))
            base = bf16_round(base)  # the 1x1 weight; the fused 2x2 taps are base/4 (exact in bf16)
            w[n + ".w"] = np.broadcast_to(base / 4.0, (cout, 2, 2, cin)).astype(np.float32).copy()
        else:
            w[n + ".w"] = _he(rng, cout, k, cin, gain)
        b = rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05)
        if n.startswith("fpn."):
            b[:] = 0.0  # DBFPN convs carry no bias / BN
        if n == "head.convt3":
            b[:] = np.float32(-0.5)
        w[n + ".b"] = b.astype(np.float32)
    if text_path:
        _install_text_path(w)
    return w


# Hand-set "code path" of the recogniser (the counterpart of the detector's text path; installed by make_rec_weights(code_path=True)).
# A seeded-random CRNN decides every time step between near-ties (top-1 / top-2 logit margins of ~0.1 sigma), so bf16 drift flips a
# few arg-maxes per line and string parity can only be bounded.  A trained network has margins of many logits; this path gives the
# seeded one the same property, through arithmetic that is EXACT (order-independent), so the engine and the oracle agree bit for bit on it:
#   rec.conv1 ch 0      : hswish(-4 * xn_G(centre tap))                         "ink" of one pixel (0 for paper, 3.4 .. 4 for ink)
#   res blocks          : carried by the residual (project row = 0)
#   b1 / b3 / b8 (the three vertical stride-2 blocks): five channels, each a chain of single-tap depthwise filters, i.e. each
#                         samples the ink at one pair of crop rows (2a, 2a + 16), a = 3 .. 7, at its own horizontal offset
#   rec.conv2, max pool : carried (single unit weights: a contraction with one non-zero product is exact in any order)
#   lstm.l0 / l1 fw 0..4: saturated gates (i = o = 1, f = 0), g = tanh(BIG * (x - thr)) -> h = +-0.7617, no recurrence;
#                         thr sits half-way between two bf16 values, so |BIG * (x - thr)| >= 16 for EVERY bf16 x
#   ctc.fc              : the 32 code classes read +-16 on the five bits: the best code leads the runner-up by >= 24 logits
#                         (dense part: sigma ~2); code 0 = CTC blank, codes 1..31 = CODE_PATH_SYMBOLS
# Every other row of every layer stays seeded-random and dense.
CODE_PATH_SYMBOLS = "etaoinshrdlucmf wypvbgkqjxz0123"          # code v (1..31) -> symbol v - 1; the space exercises word splitting
_CODE_TAPS = [(2, 4, 0), (2, 4, 1), (2, 4, 2), (3, 3, 1), (3, 3, 2)]   # (kh in b8, kh in b3, kh in b1) -> stem rows (a, a + 8), a = 3 .. 7
_CODE_KW = [(2, 2, 1), (1, 3, 0), (3, 1, 2), (2, 4, 1), (4, 2, 0)]
CODE_THR = 1.0 + 2.0 ** -8        # half-way between the bf16 neighbours 1.0 and 1.0078125: |x - thr| >= 2^-8 for every bf16 x
CODE_BIG = 4096.0                 # 2^12: |BIG * (x - thr)| >= 16


def _install_code_path(w: Dict[str, np.ndarray], num_classes: int, scale: float) -> None:
    def zero_rows(name, rows):
        for r in rows:
            w[name + ".w"][r] = 0.0
            w[name + ".b"][r] = 0.0

    bits = range(5)
    zero_rows("rec.conv1", [0])
    w["rec.conv1.w"][0, 1, 1, 1] = -4.0
    for b in rec_block_table(scale):
        p = f"rec.b{b['idx']}"
        if b["res"]:
            zero_rows(p + ".project", range(1 if b["idx"] == 0 else 5))
            continue
        stage = {1: 2, 3: 1, 8: 0}[b["idx"]]
        zero_rows(p + ".expand", bits); zero_rows(p + ".dw", bits); zero_rows(p + ".project", bits)
        for k in bits:
            w[p + ".expand.w"][k, 0, 0, 0 if b["idx"] == 1 else k] = 1.0
            w[p + ".dw.w"][k, _CODE_TAPS[k][stage], _CODE_KW[k][stage], 0] = 1.0
            w[p + ".project.w"][k, 0, 0, k] = 1.0
        if b["se"]:
            zero_rows(p + ".se2", bits)
            w[p + ".se2.b"][:5] = 4.0          # hard-sigmoid(4) = 1: the squeeze-excite gate of the path channels is exactly 1
    zero_rows("rec.conv2", bits)
    for k in bits:
        w["rec.conv2.w"][k, 0, 0, k] = 1.0
    h = REC_HIDDEN
    for layer, big, thr in ((0, CODE_BIG, CODE_THR), (1, 32.0, 0.0)):
        p = f"lstm.l{layer}.fw"
        for k in bits:
            for gate, bias in ((0, 20.0), (1, -20.0), (2, -big * thr), (3, 20.0)):   # i, f, g, o
                w[p + ".w_ih"][gate * h + k] = 0.0
                w[p + ".w_hh"][gate * h + k] = 0.0
                w[p + ".b"][gate * h + k] = bias
            w[p + ".w_ih"][2 * h + k, k] = big
    w["ctc.fc.w"][:, :5] = 0.0
    space = num_classes - 1
    for v in range(32):
        cls = 0 if v == 0 else (space if CODE_PATH_SYMBOLS[v - 1] == " " else 1 + ord(CODE_PATH_SYMBOLS[v - 1]) - 0x21)
        for k in bits:
            w["ctc.fc.w"][cls, k] = 16.0 if (v >> k) & 1 else -16.0


def make_rec_weights(seed: int = 4321, num_classes: int = 6625, scale: float = 0.5, code_path: bool = False) -> Dict[str, np.ndarray]:
    """code_path: install the hand-set exact path described above (strings then have trained-like arg-max margins)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}

    def conv(name, cout, k, cin, gain=1.0, depthwise=False):
        w[name + ".w"] = _he(rng, cout, k, 1 if depthwise else cin, gain, depthwise)
        w[name + ".b"] = (rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05)).astype(np.float32)

    c0 = rec_stem_ch(scale)
    conv("rec.conv1", c0, 3, 3)
    for b in rec_block_table(scale):
        p = f"rec.b{b['idx']}"
        hs = b["act"] == "hswish"  # hard-swish roughly halves small activations: compensate so the signal survives 11 blocks
        conv(p + ".expand", b["exp"], 1, b["cin"], gain=1.25 if hs else 1.0)
        conv(p + ".dw", b["exp"], b["k"], b["exp"], gain=1.25 if hs else 1.0, depthwise=True)
        if b["se"]:
            conv(p + ".se1", b["se_mid"], 1, b["exp"])
            conv(p + ".se2", b["exp"], 1, b["se_mid"], gain=0.7071)
        conv(p + ".project", b["cout"], 1, b["exp"], gain=(0.5 if b["res"] else 1.4) * (1.4 if b["se"] else 1.0))
    conv("rec.conv2", REC_FEAT, 1, rec_block_table(scale)[-1]["cout"], gain=1.0)
    h = REC_HIDDEN
    for layer, din in ((0, REC_FEAT), (1, 2 * h)):
        for d in ("fw", "bw"):
            p = f"lstm.l{layer}.{d}"
            w[p + ".w_ih"] = bf16_round(rng.uniform(-1, 1, (4 * h, din)).astype(np.float32) * np.float32(1.0 / np.sqrt(h)))
            w[p + ".w_hh"] = bf16_round(rng.uniform(-1, 1, (4 * h, h)).astype(np.float32) * np.float32(1.0 / np.sqrt(h)))
            w[p + ".b"] = (rng.uniform(-1, 1, 4 * h).astype(np.float32) * np.float32(1.0 / np.sqrt(h))).astype(np.float32)
    w["ctc.fc.w"] = bf16_round(rng.standard_normal((num_classes, 2 * h), dtype=np.float32) * np.float32(12.0 / np.sqrt(2 * h)))
    w["ctc.fc.b"] = (rng.standard_normal(num_classes, dtype=np.float32) * np.float32(0.1)).astype(np.float32)
    if code_path:
        _install_code_path(w, num_classes, scale)
    return w


# --------------------------------------------------------------------------------------
# Recognition, second family: SVTR-Tiny (single visual model: patch embedding -> local / global mixing blocks with height
# merging -> CTC).  BASELINE configs[4] names an SVTR recogniser with a Hindi dictionary; no implementation, weights or
# dictionary exist offline, so the architecture below is the build's restatement of the published design
# (Du et al., "SVTR: Scene Text Recognition with a Single Visual Model", tiny configuration, post-norm blocks) on the 32 x 320
# crops of this pipeline: 8 x 80 tokens -> 4 x 80 -> 2 x 80 -> 80 time steps.  "Parity unpinned".
# --------------------------------------------------------------------------------------
SVTR_VARIANTS = {
    # dims, depths, heads (head width 32), local mixing blocks (the first ones), channels of the sequence handed to the CTC head
    "tiny": dict(dims=(64, 128, 256), depths=(3, 6, 3), heads=(2, 4, 8), local_blocks=6, out=192),
    "base": dict(dims=(128, 256, 384), depths=(3, 6, 9), heads=(4, 8, 12), local_blocks=8, out=192),   # BASELINE configs[4]
}
SVTR_DIMS, SVTR_DEPTHS, SVTR_HEADS = SVTR_VARIANTS["tiny"]["dims"], SVTR_VARIANTS["tiny"]["depths"], SVTR_VARIANTS["tiny"]["heads"]
SVTR_LOCAL_BLOCKS = 6            # (tiny) the first 6 of the 12 blocks mix locally (window 7 x 11 tokens), the rest globally
SVTR_WINDOW = (7, 11)
SVTR_OUT = 192                   # channels of the sequence handed to the CTC head
SVTR_LN_EPS = 1e-6
SVTR_DTYPES = {"bf16": 0, "f16": 1}


def svtr_config(weights=None, variant: str = "tiny") -> dict:
    """The variant a weight dict / blob describes (its `svtr.config` tensor), or the named one."""
    if weights is not None and "svtr.config" in weights:
        c = [int(v) for v in np.asarray(weights["svtr.config"]).tolist()]
        return dict(dims=tuple(c[0:3]), depths=tuple(c[3:6]), heads=tuple(c[6:9]), local_blocks=c[9], out=c[10], dtype="f16" if c[11] else "bf16")
    return dict(SVTR_VARIANTS[variant], dtype="bf16")


def svtr_block_table(cfg=None) -> List[dict]:
    """One entry per mixing block: stage, dim, heads, token grid (h, w), local / global."""
    cfg = cfg or svtr_config()
    rows, idx, h = [], 0, REC_H // 4
    for s, (dim, depth, heads) in enumerate(zip(cfg["dims"], cfg["depths"], cfg["heads"])):
        for _ in range(depth):
            rows.append(dict(idx=idx, stage=s, dim=dim, heads=heads, h=h, w=REC_W // 4, local=idx < cfg["local_blocks"]))
            idx += 1
        h //= 2
    return rows


def make_svtr_weights(seed: int = 2468, num_classes: int = 6625, variant: str = "tiny", dtype: str = "bf16") -> Dict[str, np.ndarray]:
    """Seeded weights in the LOCW naming: conv / linear weights OHWI `[out, kh, kw, in]` (.w, bf16), biases and LayerNorm
    gamma / beta fp32 (.b / .g); `svtr.config` (f32 [12]) carries the variant and the storage / MFMA type the engine shall use."""
    rng = np.random.default_rng(seed)
    cfg = SVTR_VARIANTS[variant]
    dims = cfg["dims"]
    w: Dict[str, np.ndarray] = {}
    w["svtr.config"] = np.array(list(dims) + list(cfg["depths"]) + list(cfg["heads"]) + [cfg["local_blocks"], cfg["out"], SVTR_DTYPES[dtype]], np.float32)

    def conv(name, cout, k, cin, gain=1.0):
        w[name + ".w"] = _he(rng, cout, k, cin, gain)
        w[name + ".b"] = (rng.standard_normal(cout, dtype=np.float32) * np.float32(0.05)).astype(np.float32)

    def ln(name, c):
        w[name + ".g"] = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
        w[name + ".b"] = (0.05 * rng.standard_normal(c)).astype(np.float32)

    conv("svtr.pe1", dims[0] // 2, 3, 3, gain=1.3)
    conv("svtr.pe2", dims[0], 3, dims[0] // 2, gain=1.3)
    w["svtr.pos.w"] = bf16_round((0.5 * rng.standard_normal(((REC_H // 4) * (REC_W // 4), dims[0]))).astype(np.float32))
    for b in svtr_block_table(dict(cfg)):
        p, c = f"svtr.b{b['idx']}", b["dim"]
        conv(p + ".qkv", 3 * c, 1, c, gain=0.8)
        conv(p + ".proj", c, 1, c, gain=0.15)   # small residual branches: with untrained weights larger ones wash the token identity out
        ln(p + ".ln1", c)
        conv(p + ".fc1", 4 * c, 1, c, gain=1.0)
        conv(p + ".fc2", c, 1, 4 * c, gain=0.15)
        ln(p + ".ln2", c)
    for s in range(2):
        conv(f"svtr.sub{s}", dims[s + 1], 3, dims[s], gain=1.0)
        ln(f"svtr.sub{s}.ln", dims[s + 1])
    conv("svtr.last", cfg["out"], 1, dims[2], gain=1.4)
    w["svtr.ctc.fc.w"] = bf16_round(rng.standard_normal((num_classes, cfg["out"]), dtype=np.float32) * np.float32(12.0 / np.sqrt(cfg["out"])))
    w["svtr.ctc.fc.b"] = (rng.standard_normal(num_classes, dtype=np.float32) * np.float32(0.1)).astype(np.float32)
    return w


# --------------------------------------------------------------------------------------
# "LOCW" weight container
#   header : b"LOCW", u32 version(1), u32 n_tensors
#   tensor : u16 name_len, name, u8 dtype(0=f32,1=bf16), u8 ndim, u32 dims[ndim], u64 nbytes,
#            pad to 16-byte file offset, raw data
# --------------------------------------------------------------------------------------
BLOB_MAGIC = b"LOCW"


def write_blob(weights: Dict[str, np.ndarray]) -> bytes:
    out = bytearray()
    out += BLOB_MAGIC + struct.pack("<II", 1, len(weights))
    for name, arr in weights.items():
        nb = name.encode()
        as_bf16 = name.endswith(".w") or name.endswith(".w_ih") or name.endswith(".w_hh")
        data = f32_to_bf16_bits(arr).tobytes() if as_bf16 else np.ascontiguousarray(arr, np.float32).tobytes()
        out += struct.pack("<H", len(nb)) + nb + struct.pack("<BB", 1 if as_bf16 else 0, arr.ndim)
        out += struct.pack("<%dI" % arr.ndim, *arr.shape) + struct.pack("<Q", len(data))
        out += b"\x00" * ((-len(out)) % 16)
        out += data
    return bytes(out)


def read_blob(blob: bytes) -> Dict[str, np.ndarray]:
    assert blob[:4] == BLOB_MAGIC, "not a LOCW blob"
    ver, n = struct.unpack_from("<II", blob, 4)
    assert ver == 1
    off = 12
    out: Dict[str, np.ndarray] = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", blob, off); off += 2
        name = blob[off:off + ln].decode(); off += ln
        dt, nd = struct.unpack_from("<BB", blob, off); off += 2
        dims = struct.unpack_from("<%dI" % nd, blob, off); off += 4 * nd
        (nbytes,) = struct.unpack_from("<Q", blob, off); off += 8
        off += (-off) % 16
        raw = blob[off:off + nbytes]; off += nbytes
        if dt == 1:
            out[name] = bf16_bits_to_f32(np.frombuffer(raw, np.uint16).reshape(dims))
        else:
            out[name] = np.frombuffer(raw, np.float32).reshape(dims).copy()
    return out
