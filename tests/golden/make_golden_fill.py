"""Golden vectors for the unused-by-the-driver hole filler `dibr_filter_mask` (utils.py:345-392), produced by IMPORTING the reference in
this container (run from /root/repo: python tests/golden/make_golden_fill.py). Inputs: the hole-punched merged warp of warp.npz (the same
inputs `dibr_filter_mask2`'s golden uses) and a second, sparser map that exercises the second pass, the border fills and the erase pass."""
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, "/root/reference")


def inputs():
    gw = np.load(os.path.join(HERE, "warp.npz"))
    known = gw["fill_in_mask"].astype(np.int64)
    img = gw["warp_image"].copy()
    img[(known == 0) & (gw["warp_mask"] == 1)] = 1.0
    # second case: a seeded random map (55 % known) over the same picture, borders included
    g = np.random.Generator(np.random.PCG64(43))
    known2 = (g.uniform(0, 1, known.shape) < 0.55).astype(np.int64)
    img2 = gw["warp_image"].copy()
    img2[known2 == 0] = 1.0
    return (img, known), (img2, known2)


def main():
    from unittest.mock import MagicMock
    for name in ["cv2", "imageio", "imageio.v2", "torchvision", "torchvision.transforms", "statsmodels", "statsmodels.api",
                 "skimage", "skimage.io", "skimage.metrics", "skimage.measure", "lpips", "plyfile", "kornia", "configargparse"]:
        sys.modules.setdefault(name, MagicMock())   # inert stubs: the function touches none of them (as in make_golden_warp.py)
    import utils as ref_utils
    out = {}
    for tag, (img, known) in zip(("a", "b"), inputs()):
        f_img, f_map = ref_utils.dibr_filter_mask(img.copy(), known.copy())
        out[f"fill1_{tag}_image"], out[f"fill1_{tag}_mask"] = f_img, f_map.astype(np.uint8)
        print(tag, "known before", int(known.sum()), "after", int(f_map.sum()), "pixels set to 255:", int((f_img == 255).all(-1).sum()))
    np.savez_compressed(os.path.join(HERE, "fill.npz"), **out)


if __name__ == "__main__":
    main()
