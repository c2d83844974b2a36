"""Flow file I/O and end-point-error metrics for the harness: what is needed to
score a flow field against Sintel / KITTI ground truth once data is available on the box.
Formats as the reference reads and writes them:
  .flo  Middlebury: float32 tag 202021.25, int32 width, int32 height, then (h, w, 2) float32 (u, v)
        interleaved, little endian (core/utils/frame_utils.py:12-31, :70-99)
  KITTI 16-bit RGB PNG: R = 64 u + 2^15, G = 64 v + 2^15, B = valid (frame_utils.py:102-107, :116-120).
        The reference goes through OpenCV, which is not installed here; 48-bit PNG is decoded /
        encoded directly (zlib + the five PNG row filters), nothing else of PNG is needed.
Metrics: core/evaluate.py:131-145 (EPE, 1/3/5 px), :276-298 (KITTI EPE and F1-all)."""
import struct
import zlib

import numpy as np

FLO_TAG = 202021.25


def read_flo(path):
    with open(path, "rb") as f:
        tag, = struct.unpack("<f", f.read(4))
        if tag != FLO_TAG:
            raise ValueError(f"{path}: not a .flo file (tag {tag})")
        w, h = struct.unpack("<ii", f.read(8))
        data = np.frombuffer(f.read(8 * w * h), dtype="<f4")
    if data.size != 2 * w * h:
        raise ValueError(f"{path}: truncated .flo file")
    return data.reshape(h, w, 2).astype(np.float32)


def write_flo(path, flow):
    flow = np.asarray(flow, dtype="<f4")
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError("flow must be (H, W, 2)")
    with open(path, "wb") as f:
        f.write(struct.pack("<fii", FLO_TAG, flow.shape[1], flow.shape[0]))
        f.write(np.ascontiguousarray(flow).tobytes())


# ---- 16-bit RGB PNG ------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)).astype(np.uint8)


def read_png16_rgb(path):
    """(H, W, 3) uint16 from a non-interlaced 16-bit truecolour PNG."""
    raw = open(path, "rb").read()
    if raw[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        n, kind = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if (depth, ctype, interlace) != (16, 2, 0):
        raise ValueError(f"{path}: expected 16-bit RGB, non-interlaced (got depth {depth}, type {ctype})")
    bpp, stride = 6, 6 * w
    data = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    for y in range(h):
        ft, line = int(data[y, 0]), data[y, 1:].copy()
        if ft in (1, 3, 4):                                    # depend on the reconstructed left pixel
            for x in range(0, stride, bpp):
                left = line[x - bpp:x] if x else np.zeros(bpp, np.uint8)
                up = prev[x:x + bpp]
                if ft == 1:
                    line[x:x + bpp] += left
                elif ft == 3:
                    line[x:x + bpp] += ((left.astype(np.int32) + up) // 2).astype(np.uint8)
                else:
                    ul = prev[x - bpp:x] if x else np.zeros(bpp, np.uint8)
                    line[x:x + bpp] += _paeth(left, up, ul)
        elif ft == 2:
            line += prev
        elif ft != 0:
            raise ValueError(f"{path}: bad PNG filter {ft}")
        out[y] = prev = line
    return out.view(">u2").reshape(h, w, 3).astype(np.uint16)


def write_png16_rgb(path, img):
    img = np.asarray(img)
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("image must be (H, W, 3)")
    h, w, _ = img.shape
    rows = np.concatenate([np.zeros((h, 1), np.uint8), img.astype(">u2").view(np.uint8).reshape(h, 6 * w)], axis=1)

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + chunk(b"IEND", b""))


def read_flow_kitti(path):
    """-> flow (H, W, 2) float32, valid (H, W) float32"""
    rgb = read_png16_rgb(path).astype(np.float32)
    return (rgb[..., :2] - 2 ** 15) / 64.0, rgb[..., 2]


def write_flow_kitti(path, flow, valid=None):
    flow = np.asarray(flow, dtype=np.float64)
    valid = np.ones(flow.shape[:2]) if valid is None else np.asarray(valid)
    uv = np.clip(64.0 * flow + 2 ** 15, 0, 65535)
    write_png16_rgb(path, np.concatenate([uv, valid[..., None]], axis=-1).astype(np.uint16))


# ---- metrics -------------------------------------------------------------------------------------
def epe_map(flow, flow_gt):
    """per-pixel end-point error; flow arrays (..., 2) or (2, H, W) pairs of equal shape"""
    flow, flow_gt = np.asarray(flow, np.float64), np.asarray(flow_gt, np.float64)
    axis = -1 if flow.shape[-1] == 2 else 0
    return np.sqrt(((flow - flow_gt) ** 2).sum(axis=axis))


def sintel_metrics(epes):
    """epes: list of per-frame EPE maps -> EPE and 1/3/5 px inlier rates over all pixels"""
    e = np.concatenate([np.ravel(x) for x in epes])
    return {"epe": float(e.mean()), "1px": float((e < 1).mean()), "3px": float((e < 3).mean()),
            "5px": float((e < 5).mean())}


def kitti_metrics(flows, flows_gt, valids):
    """KITTI-15: mean of per-frame EPE over valid pixels; F1-all = % of valid pixels with
    EPE > 3 px and EPE > 5 % of the ground-truth magnitude"""
    per_frame, outliers = [], []
    for fl, gt, va in zip(flows, flows_gt, valids):
        e = epe_map(fl, gt).ravel()
        axis = -1 if np.asarray(gt).shape[-1] == 2 else 0
        mag = np.sqrt((np.asarray(gt, np.float64) ** 2).sum(axis=axis)).ravel()
        v = np.asarray(va).ravel() >= 0.5
        per_frame.append(e[v].mean())
        with np.errstate(divide="ignore", invalid="ignore"):
            outliers.append(((e > 3.0) & (e / mag > 0.05))[v])
    return {"kitti-epe": float(np.mean(per_frame)), "kitti-f1": float(100 * np.concatenate(outliers).mean())}
