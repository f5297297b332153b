"""ScreenGrabber::saveRenderTargetToFile (what Environment::saveImage calls): the built-in PNG encoder -- fixed-Huffman
deflate with previous-pixel matches -- must produce a file that a standard zlib inflates to the frame it was given.  Host
code only: runs without a GPU."""
import os
import subprocess

import numpy as np

from test_gpu_render import read_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_png_writer_round_trips_through_zlib(ok, tmp_path):
    exe, png = str(tmp_path / "png_check"), str(tmp_path / "frame.png")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "png_check.cpp"),
                    "-L", os.path.join(ROOT, "openkitchen_amd"), "-lokenv", "-Wl,-rpath," + os.path.join(ROOT, "openkitchen_amd")], check=True)
    subprocess.run([exe, png], check=True, timeout=120)
    img = read_png(png)
    y, x = np.mgrid[0:1400, 0:1600]
    assert (img[..., 0] == (x // 100) * 16).all()
    assert (img[..., 1] == np.where(y > 700, 255, (x * 7 + y * 3) % 251)).all()
    assert (img[..., 2] == np.where(x == y, 255, 0)).all() and (img[..., 3] == 255).all()
    assert os.path.getsize(png) < 1600 * 1400 * 4 // 2  # the flat half compresses away
