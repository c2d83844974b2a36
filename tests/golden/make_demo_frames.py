#!/usr/bin/env python3
"""Fixture generator (runs only where /root/reference exists): the demo frame pair that
demo.py of the reference is run on (demo-frames/frame_0016.png, frame_0017.png; 436 x 1024 RGB,
BASELINE.json configs[0]) decoded to uint8 arrays -> tests/golden/demo_frames.npz.
Data only: no code of the reference is read or stored."""
import os
import numpy as np
from PIL import Image

SRC = "/root/reference/demo-frames"
HERE = os.path.dirname(os.path.abspath(__file__))
f1 = np.array(Image.open(os.path.join(SRC, "frame_0016.png")).convert("RGB"))
f2 = np.array(Image.open(os.path.join(SRC, "frame_0017.png")).convert("RGB"))
assert f1.shape == f2.shape == (436, 1024, 3) and f1.dtype == np.uint8
np.savez_compressed(os.path.join(HERE, "demo_frames.npz"), frame1=f1, frame2=f2)
print("wrote demo_frames.npz", f1.shape)
