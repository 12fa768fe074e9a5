"""Golden vectors for the front-end glue of the reference (build container only; test infrastructure).

The reference's evaluation.py cannot be imported here (cPickle, lsdpython, skimage, scipy.ndimage.imread are absent),
but two of its functions are plain NumPy around those imports: ``detect_lsd_lines`` (evaluation.py:227-251: pixel ->
normalised image coordinates, y up, long side = [-1, 1]) and ``create_data_dict_single`` (:188-224: homogeneous lines
= cross(p1, p2)).  Their source text is read from /root/reference, compiled IN MEMORY (print statements fixed like
ref_shim does) and run with stand-ins for the absent imports -- an `lsd` whose detect_line_segments returns a seeded
N x 7 array, a `color.rgb2gray` that applies skimage's documented weights, a `get_sphere_image` that returns None --
so what is captured is exactly the reference's own arithmetic on known detector output.  Only data is written:
tests/golden/frontend.npz.
"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from ref_shim import REFERENCE_ROOT, _fix_print  # noqa: E402


def load_functions(names):
    with open(os.path.join(REFERENCE_ROOT, "evaluation.py")) as fh:
        src = _fix_print(fh.read(), "evaluation")
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names)
    mod = ast.Module(body=keep, type_ignores=[])
    env = {"np": np, "print": lambda *a, **k: None}
    exec(compile(mod, "evaluation.py", "exec"), env)
    return env


def main():
    rs = np.random.RandomState(11)
    env = load_functions(["detect_lsd_lines", "create_data_dict_single"])
    out = {}
    for case, (h, w, n) in enumerate([(480, 640, 37), (1333, 2000, 211), (800, 533, 5)]):
        seg = np.zeros((n, 7))
        seg[:, 0], seg[:, 2] = rs.uniform(0, w, n), rs.uniform(0, w, n)
        seg[:, 1], seg[:, 3] = rs.uniform(0, h, n), rs.uniform(0, h, n)
        seg[:, 4], seg[:, 5], seg[:, 6] = rs.uniform(1, 4, n), 0.125, rs.uniform(0.1, 50, n)
        env["lsd"] = types.SimpleNamespace(detect_line_segments=lambda image, seg=seg: seg.copy())
        env["color"] = types.SimpleNamespace(
            rgb2gray=lambda rgb: (rgb.astype(np.float64) / 255.0).dot(np.array([0.2125, 0.7154, 0.0721])))
        env["get_sphere_image"] = lambda lines, size=250, alpha=0.1: None
        gray = rs.uniform(0, 1, (h, w))
        res = env["detect_lsd_lines"](gray.copy())
        rgb = rs.randint(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
        # create_data_dict_single sees the detector through detect_lsd_lines; the image only sets the normalisation
        seg_small = seg.copy()
        seg_small[:, 0:4] /= 8.0
        env["lsd"] = types.SimpleNamespace(detect_line_segments=lambda image, s=seg_small: s.copy())
        single = env["create_data_dict_single"](rgb, 500)
        out.update({"shape%d" % case: np.array([h, w]), "raw%d" % case: seg, "segments%d" % case: res["segments"],
                    "nfa%d" % case: res["nfa"], "rgb%d" % case: rgb, "raw_small%d" % case: seg_small,
                    "single_segments%d" % case: single["lines"]["line_segments"], "single_lines%d" % case: single["lines"]["lines"],
                    "single_shape%d" % case: np.array(single["lines"]["image_shape"])})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frontend.npz"), **out)
    print("frontend golden written:", sorted(out)[:6], "...")


if __name__ == "__main__":
    main()
