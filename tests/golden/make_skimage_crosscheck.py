#!/opt/conda/bin/python3.9
"""Cross-check vectors from the one INDEPENDENT implementation of this algorithm family in the build image:
scikit-image 0.18.3 (BSD) under /opt/conda/bin/python3.9 -- not the reference, not OpenCV, never shipped to the GPU box.

  /opt/conda/bin/python3.9 tests/golden/make_skimage_crosscheck.py     ->  tests/golden/skimage_crosscheck.npz

It pins nothing about OpenCV 2.4.13 (the reference's ORB/BFMatcher); it is the only non-self-referential evidence for
two definitions the oracle restates from memory:
  * FAST-9/16 segment test, threshold 20: skimage.feature.corner_fast(img, n=9, threshold=20.5/255) != 0 is the set of
    pixels with >= 9 contiguous ring pixels all brighter than I+20 or all darker than I-20 (strict, like OpenCV's
    `> I + t`: on u8 data d > 20.5 <=> d >= 21), before any non-max suppression;
  * brute-force Hamming nearest neighbour with ties to the lower train index:
    skimage.feature.match_descriptors(a, b, metric='hamming', cross_check=False) (argmin of cdist).
Inputs and outputs are both stored, so the consuming test (tests/test_oracle_cpu.py) needs neither this interpreter nor
scikit-image."""
import os
import sys

import numpy as np
from skimage.feature import corner_fast, match_descriptors

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from monocular_slam_amd import synth  # noqa: E402  (pure numpy generators; whatever they give here is stored)

out = {}
for seed in (11, 12):
    img = synth.textured_frame(240, 180, seed)
    resp = corner_fast(img, n=9, threshold=20.5 / 255.0)
    out[f"img_{seed}"] = img
    out[f"fast_mask_{seed}"] = np.packbits(resp != 0)
for seed in (21, 22):
    a, b, _ = synth.descriptor_sets(300, seed, flip_p=0.1, junk_frac=0.25)
    b[5] = b[3]; b[200] = b[100]  # exact duplicates in the train set: the tie rule decides
    m = match_descriptors(np.unpackbits(a, axis=1).astype(bool), np.unpackbits(b, axis=1).astype(bool), metric="hamming",
                          cross_check=False)
    assert (m[:, 0] == np.arange(300)).all()
    out[f"desc_a_{seed}"], out[f"desc_b_{seed}"], out[f"nn_{seed}"] = a, b, m[:, 1].astype(np.int32)
np.savez_compressed(os.path.join(HERE, "skimage_crosscheck.npz"), **out)
print("wrote skimage_crosscheck.npz:", {k: v.shape for k, v in out.items()})
