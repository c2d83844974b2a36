"""harness/flow_io.py: Middlebury .flo and KITTI 16-bit PNG flow files, EPE metrics (CPU)."""
import struct
import zlib

import numpy as np
import pytest

from harness import flow_io as fio


def test_flo_round_trip_and_layout(tmp_path):
    rs = np.random.RandomState(0)
    flow = rs.randn(7, 11, 2).astype(np.float32) * 20
    p = tmp_path / "a.flo"
    fio.write_flo(p, flow)
    raw = p.read_bytes()
    assert struct.unpack("<fii", raw[:12]) == (202021.25, 11, 7)          # tag, width, height
    assert np.frombuffer(raw[12:20], "<f4").tolist() == flow[0, 0].tolist()   # (u, v) interleaved, row-major
    assert np.array_equal(fio.read_flo(p), flow)
    p.write_bytes(b"\x00" * 20)
    with pytest.raises(ValueError):
        fio.read_flo(p)


def test_kitti_png_round_trip(tmp_path):
    rs = np.random.RandomState(1)
    flow = np.round(rs.randn(9, 13, 2) * 30 * 64) / 64            # representable: 1/64 px steps
    valid = (rs.rand(9, 13) > 0.3).astype(np.float32)
    p = tmp_path / "k.png"
    fio.write_flow_kitti(p, flow, valid)
    f2, v2 = fio.read_flow_kitti(p)
    assert np.array_equal(f2, flow.astype(np.float32)) and np.array_equal(v2, valid)
    from PIL import Image                                         # an independent decoder accepts the file
    assert Image.open(p).size == (13, 9)


@pytest.mark.parametrize("ftype", [1, 2, 3, 4])
def test_png_decoder_handles_every_row_filter(tmp_path, ftype):
    """Files written by other encoders use the Sub/Up/Average/Paeth filters: re-encode rows by hand."""
    rs = np.random.RandomState(ftype)
    img = rs.randint(0, 65536, (5, 4, 3)).astype(np.uint16)
    rows = img.astype(">u2").view(np.uint8).reshape(5, 24).astype(np.int32)
    enc = np.zeros((5, 25), np.uint8)
    for y in range(5):
        up = rows[y - 1] if y else np.zeros(24, np.int32)
        left = np.concatenate([np.zeros(6, np.int32), rows[y, :-6]])
        ul = np.concatenate([np.zeros(6, np.int32), up[:-6]])
        if ftype == 1:
            pred = left
        elif ftype == 2:
            pred = up
        elif ftype == 3:
            pred = (left + up) // 2
        else:
            p = left + up - ul
            pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        enc[y, 0] = ftype
        enc[y, 1:] = (rows[y] - pred) % 256

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    path = tmp_path / "f.png"
    path.write_bytes(fio._PNG_SIG +
                     chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 5, 16, 2, 0, 0, 0)) +
                     chunk(b"IDAT", zlib.compress(enc.tobytes())) + chunk(b"IEND", b""))
    assert np.array_equal(fio.read_png16_rgb(path), img)


def test_metrics_known_answers():
    gt = np.zeros((2, 4, 2)); gt[..., 0] = 10.0
    fl = gt.copy(); fl[0, 0] = [13.0, 4.0]                        # one pixel off by (3, 4): EPE 5
    e = fio.epe_map(fl, gt)
    assert e[0, 0] == 5.0 and e.sum() == 5.0
    m = fio.sintel_metrics([e])
    assert m["epe"] == 5.0 / 8 and m["1px"] == 7 / 8 and m["5px"] == 7 / 8
    k = fio.kitti_metrics([fl], [gt], [np.ones((2, 4))])
    assert abs(k["kitti-epe"] - 5.0 / 8) < 1e-12 and abs(k["kitti-f1"] - 100 / 8) < 1e-12   # 5 px > 3 px and > 5 % of 10
    k = fio.kitti_metrics([fl], [gt], [np.concatenate([np.zeros((1, 4)), np.ones((1, 4))])])
    assert k == {"kitti-epe": 0.0, "kitti-f1": 0.0}               # the bad pixel is not valid


def test_reference_checkpoint_with_dataparallel_prefix_loads():
    """core/demo.py:43-46 loads `module.`-prefixed state dicts (saved from nn.DataParallel,
    core/train.py:342,398-400); RAFT.load_reference_checkpoint must accept them unchanged, and a
    foreign key must still be an error (strict)."""
    import pytest
    import torch
    from harness.raft_model import RAFT
    torch.manual_seed(3)
    src = RAFT(lambda a, b: None)
    sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    dst = RAFT(lambda a, b: None)
    res = dst.load_reference_checkpoint(sd)
    assert not res.missing_keys and not res.unexpected_keys
    for (ka, va), (kb, vb) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert sum(p.numel() for p in dst.parameters()) == 5_257_536          # RAFT-basic, SURVEY.md 0.1
    plain = dict(src.state_dict())                                          # unprefixed dicts load too
    RAFT(lambda a, b: None).load_reference_checkpoint(plain)
    sd["module.not_a_parameter"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        RAFT(lambda a, b: None).load_reference_checkpoint(sd)
