"""Front end of the pipeline: image -> grey levels -> line segments in normalised image
coordinates -> homogeneous lines -> sphere raster.  Mirrors evaluation.py:121-251 of the reference.

What can be pinned is pinned: `detect_lsd_lines` (the pixel -> normalised-coordinate arithmetic around the detector)
and the homogeneous lines of `create_data_dict_single` are checked against the reference's own functions run on known
detector output (tests/golden/frontend.npz, tests/test_frontend.py).  What cannot: the detector itself (lsd.py; the
reference's is an absent submodule), skimage's rgb2gray (absent here; its documented weights are used) and
ImageMagick's `convert -resize` (an external program in the reference, evaluation.py:142-143; Pillow's Lanczos
resampling stands in).  The raster comes from the GPU rasteriser like everywhere else in this package."""
import numpy as np

from . import lsd

_GRAY = np.array([0.2125, 0.7154, 0.0721])     # skimage.color.rgb2gray (evaluation.py:150,190)


def rgb2gray(image_rgb):
    """float64 luminance in [0, 1] for uint8 input, like skimage.color.rgb2gray."""
    a = np.asarray(image_rgb)
    if a.ndim == 2:
        return a.astype(np.float64) / (255.0 if a.dtype == np.uint8 else 1.0)
    f = a[..., :3].astype(np.float64)
    if a.dtype == np.uint8:
        f /= 255.0
    return f.dot(_GRAY)


def imread(path):
    """scipy.ndimage.imread (evaluation.py:145,148): the file as an array, RGB for colour images."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB"):
            im = im.convert("RGB")
        return np.asarray(im).copy()


def resize_to_fit(image, target_size):
    """`convert file -resize SxS` (evaluation.py:142-143): scale to fit inside S x S keeping the aspect ratio."""
    from PIL import Image
    h, w = image.shape[:2]
    s = min(float(target_size) / w, float(target_size) / h)
    nw, nh = max(1, int(np.floor(w * s + 0.5))), max(1, int(np.floor(h * s + 0.5)))
    im = Image.fromarray(image)
    return np.asarray(im.resize((nw, nh), Image.LANCZOS)).copy()


def detect_lsd_lines(image, detector=None):
    """evaluation.py:227-251: grey image -> {'segments': N x 4 in normalised coordinates, 'nfa': N}.
    x, y are centred, divided by half of the LONG side and y points up."""
    image = np.asarray(image).astype('float64')
    if np.max(image) <= 1:
        image = image * 255
    width = image.shape[1]
    height = image.shape[0]
    scale_w = np.maximum(width, height)
    scale_h = scale_w
    lsd_lines = np.array((detector or lsd.detect_line_segments)(image), dtype=np.float64)
    lsd_lines = lsd_lines.reshape(-1, 7)
    lsd_lines[:, 0] -= width / 2.0
    lsd_lines[:, 1] -= height / 2.0
    lsd_lines[:, 2] -= width / 2.0
    lsd_lines[:, 3] -= height / 2.0
    lsd_lines[:, 0] /= (scale_w / 2.0)
    lsd_lines[:, 1] /= (scale_h / 2.0)
    lsd_lines[:, 2] /= (scale_w / 2.0)
    lsd_lines[:, 3] /= (scale_h / 2.0)
    lsd_lines[:, 1] *= -1
    lsd_lines[:, 3] *= -1
    return {'segments': lsd_lines[:, 0:4], 'nfa': lsd_lines[:, 6]}


def homogeneous_lines(line_segments):
    """evaluation.py:161-168 / :199-208: l = cross((x1, y1, 1), (x2, y2, 1)) per segment."""
    seg = np.asarray(line_segments, dtype=np.float64).reshape(-1, 4)
    p1 = np.concatenate([seg[:, 0:2], np.ones((seg.shape[0], 1))], 1)
    p2 = np.concatenate([seg[:, 2:4], np.ones((seg.shape[0], 1))], 1)
    return np.cross(p1, p2)


def create_data_dict_single(image_rgb, cnn_input_size=250, detector=None, sphere_fn=None):
    """evaluation.py:188-224: one image -> {'lines': {...}, 'sphere_image': raster}."""
    image = rgb2gray(image_rgb)
    datum = {"image_shape": image.shape, "image": image_rgb}
    lsd_result = detect_lsd_lines(image, detector)
    datum['line_segments'] = lsd_result['segments']
    datum['lines'] = homogeneous_lines(lsd_result['segments'])
    if sphere_fn is None:
        from .evaluation import get_sphere_image as sphere_fn
    return {'lines': datum, 'sphere_image': sphere_fn(datum['lines'], size=cnn_input_size, alpha=0.1)}


def line_detector(image_file, target_size=None):
    """The callable evaluation.create_data_pickles plugs in: (image_file, target_size) -> (image_rgb, segments)."""
    image_rgb = imread(image_file)
    if target_size is not None:
        image_rgb = resize_to_fit(image_rgb, target_size)
    return image_rgb, detect_lsd_lines(rgb2gray(image_rgb))['segments']
