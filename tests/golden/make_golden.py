"""Derives the committed golden fixtures from the reference's own data files.

Run once in the build container (it reads /root/reference, which does not exist on the
GPU box):  python tests/golden/make_golden.py

  output_stride4.npy  every 4th pixel (x and y) of the reference's rendered image
                      output.png, RGB, uint8 [270, 480, 3]
  output_meta.json    size, sha256 of the file and of the decoded RGBA bytes, spot pixels
  test01.yaml         the reference's only scene file (assets/scenes/test01.yaml), the
                      input that output.png was rendered from (src/main.rs:17-20)
"""
import hashlib
import json
import os
import shutil

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    path = os.path.join(REF, "output.png")
    raw = open(path, "rb").read()
    img = np.asarray(Image.open(path))
    assert img.shape == (1080, 1920, 4) and img.dtype == np.uint8
    np.save(os.path.join(HERE, "output_stride4.npy"), np.ascontiguousarray(img[::4, ::4, :3]))
    spots = [(960, 540), (800, 400), (690, 435), (385, 820), (1500, 740), (100, 700), (960, 900), (1200, 700),
             (1919, 1079), (0, 1079), (960, 374), (960, 380), (960, 1000), (0, 0)]
    meta = {
        "source": "output.png of mattkleiny/raytracer-rust v1 (rendered by src/main.rs:17-20)",
        "width": 1920,
        "height": 1080,
        "file_sha256": hashlib.sha256(raw).hexdigest(),
        "decoded_rgba_sha256": hashlib.sha256(img.tobytes()).hexdigest(),
        "alpha_all_255": bool((img[..., 3] == 255).all()),
        "black_fraction": float((img[..., :3].max(axis=2) == 0).mean()),
        "mean_rgb": [float(img[..., c].mean()) for c in range(3)],
        "spots": [{"x": x, "y": y, "rgb": [int(v) for v in img[y, x, :3]]} for x, y in spots],
        "row_sha256_every_135": {str(y): hashlib.sha256(img[y].tobytes()).hexdigest() for y in range(0, 1080, 135)},
    }
    with open(os.path.join(HERE, "output_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.copyfile(os.path.join(REF, "assets", "scenes", "test01.yaml"), os.path.join(HERE, "test01.yaml"))
    print(json.dumps(meta, indent=1)[:600])


if __name__ == "__main__":
    main()
