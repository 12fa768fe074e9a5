"""The reference's evaluation.py call surface (evaluation.py:12-361) on the MI355X path.

Same function names, argument order and defaults; the per-image pickle schema is unchanged
({'lines': {...,'line_segments','lines'}, 'sphere_image', 'cnn_prediction', 'EM_result'}).  What
changes is behind the calls: run_cnn pushes ALL rasters of the dataset through the batched HIP CNN
(the reference loops caffe_forward with batch 1, :271-290) and run_em refines ALL images in one
vpk_em_batch launch (the reference loops run_em_single, :309-329)."""
import glob
import os
import pickle
import time

import numpy as np

from . import caffe_io, cnn, em as _em, sphere_mapping, vp_localisation as vp

PICKLE_PROTOCOL = 2      # readable by the reference's Python 2 (it writes protocol -1 of cPickle)


class _ReferenceUnpickler(pickle.Unpickler):
    """The reference's scripts run with their directory on sys.path, so its pickles name the module
    ``probability_functions`` at top level; resolve it to this package's record types."""

    def find_class(self, module, name):
        if module == "probability_functions":
            from . import probability_functions as pf
            return getattr(pf, name)
        return super().find_class(module, name)


def _load_pickle(path):
    with open(path, 'rb') as fp:
        try:
            return _ReferenceUnpickler(fp).load()
        except UnicodeDecodeError:
            fp.seek(0)
            return _ReferenceUnpickler(fp, encoding='latin1').load()   # written by the Python 2 reference


def _dump_pickle(obj, path):
    with open(path, 'wb') as fp:
        pickle.dump(obj, fp, PICKLE_PROTOCOL)


def get_sphere_image(lines, size=250, alpha=0.1, f=1.0):
    """evaluation.py:12-14."""
    return sphere_mapping.sphere_line_plot(lines, size, alpha=alpha, f=f, alternative=False)


def init_caffe(model_def, model_weights, gpu_id=0, mean_file=None):
    """evaluation.py:17-22 -> a cnn.Net on GPU gpu_id.  The mean blob is fused into conv1's load, so
    it is bound here (mean_file) or on the first caffe_forward(net, image, mean_arr) call."""
    if model_def and os.path.isfile(model_def):
        caffe_io.check_deploy_prototxt(model_def)
    layers = caffe_io.read_caffemodel(model_weights)
    weights = {}
    for name, _ in cnn.LAYER_SHAPES:
        key = "fc8_20x20" if name == "fc8" else name
        if key not in layers or len(layers[key]) < 2:
            raise ValueError("caffemodel %s lacks layer %s" % (model_weights, key))
        w, b = layers[key][0], layers[key][1]
        weights[name] = (w.reshape(dict(cnn.LAYER_SHAPES)[name]), b.reshape(-1))
    return cnn.LazyNet(weights, device=gpu_id, mean=None if mean_file is None else read_mean_blob(mean_file))


def read_mean_blob(mean_file):
    """evaluation.py:25-31 -> ndarray (1, 1, 500, 500)."""
    return caffe_io.read_binaryproto(mean_file)


def caffe_forward(net, image, mean_arr):
    """evaluation.py:34-38: uint8 500x500 raster -> (20, 20) float32 sigout."""
    return net.forward_single(image, mean_arr)


def get_data_list(source_folder, destination_folder, name, cnn_model_root, cnn_model_iterations,
                  dataset_name=None, distance_measure="angle", use_weights=True, do_split=True, do_merge=True,
                  update=False):
    """evaluation.py:55-118 (file discovery only; unchanged behaviour, plain host code)."""
    tag = "%s_%s_%sweights_%ssplit_%smerge" % (name, distance_measure, "" if use_weights else "no",
                                               "" if do_split else "no", "" if do_merge else "no")
    pkl_filename = "%s/%s.pkl" % (destination_folder, tag)
    if os.path.isfile(pkl_filename) and not update:
        return _load_pickle(pkl_filename)
    dataset = {'source_folder': source_folder, 'destination_folder': destination_folder + '/' + tag,
               'cnn_root': cnn_model_root, 'cnn_iterations': cnn_model_iterations, 'use_weights': use_weights,
               'distance_measure': distance_measure, 'do_split': do_split, 'do_merge': do_merge}
    os.makedirs(dataset['destination_folder'], exist_ok=True)
    if dataset_name == 'york':
        image_files = glob.glob("%s/P*/P*.jpg" % source_folder)
    elif dataset_name == 'horizon':
        with open("%s/split/test.txt" % source_folder) as fp:
            image_files = ["%s/images/%s" % (source_folder, ln.strip()) for ln in fp if ln.strip()]
    elif dataset_name == 'eurasian':
        image_files = glob.glob("%s/*.jpg" % source_folder)
    else:
        image_files = sum((glob.glob("%s/*.%s" % (source_folder, ext)) for ext in ("jpg", "png", "pgm")), [])
    image_files.sort()
    dataset['image_files'] = image_files
    dataset['pickle_files'] = ["%s/%s.data.pkl" % (dataset["destination_folder"],
                                                  os.path.splitext(os.path.basename(f))[0]) for f in image_files]
    dataset['name'] = tag
    _dump_pickle(dataset, pkl_filename)
    return dataset


def create_data_pickles(dataset, update=False, cnn_input_size=250, target_size=None, line_detector=None):
    """evaluation.py:121-186.  ``line_detector`` (image_file, target_size) -> (image_rgb, segments N x 4 in the
    reference's normalised coordinates) defaults to this package's front end (frontend.line_detector: the
    reference's `lsdpython` submodule is empty and ImageMagick is an external program, so both are stand-ins --
    frontend.py says what is pinned and what is not); the pickles are the reference's, rasterised on the GPU."""
    if line_detector is None:       # this package's own front end (frontend.py: Pillow + the LSD of csrc/vpk_lsd.cpp)
        from . import frontend
        line_detector = frontend.line_detector
    for image_file, data_file in zip(dataset["image_files"], dataset["pickle_files"]):
        if os.path.isfile(data_file) and not update:
            continue
        image_rgb, segs = line_detector(image_file, target_size)
        segs = np.ascontiguousarray(segs, dtype=np.float64)
        p1 = np.concatenate([segs[:, 0:2], np.ones((segs.shape[0], 1))], 1)
        p2 = np.concatenate([segs[:, 2:4], np.ones((segs.shape[0], 1))], 1)
        lines = np.cross(p1, p2)                                             # :161-168
        datum = {"dataset": dataset["name"], "image_file": image_file, "image_shape": image_rgb.shape[:2],
                 "image": image_rgb, "line_segments": segs, "lines": lines}
        sphere_image = get_sphere_image(datum['lines'], size=cnn_input_size, alpha=0.1)   # :175
        _dump_pickle({'lines': datum, 'sphere_image': sphere_image}, data_file)


def create_data_dict_single(image_rgb, cnn_input_size=250):
    """evaluation.py:188-224."""
    from . import frontend
    return frontend.create_data_dict_single(image_rgb, cnn_input_size)


def detect_lsd_lines(image):
    """evaluation.py:227-251."""
    from . import frontend
    return frontend.detect_lsd_lines(image)


def run_cnn(dataset, model_def, model_weights, mean_file, gpu=0, net=None):
    """evaluation.py:254-292, batched: every raster of the dataset goes through one forward call."""
    start = time.time()
    mean_arr = read_mean_blob(mean_file) if net is None else None
    if net is None:
        net = init_caffe(model_def, model_weights, gpu)
    print("CNN init time: ", time.time() - start)
    files = [f[0] if isinstance(f, tuple) else f for f in dataset['pickle_files']]
    data = [_load_pickle(f) for f in files]
    idx = [i for i, d in enumerate(data) if d['sphere_image'] is not None]
    if idx:
        sphere = np.stack([data[i]['sphere_image'] for i in idx])
        pred = net.forward_batch(sphere, mean_arr)
        for k, i in enumerate(idx):
            data[i]['cnn_prediction'] = pred[k]
    for i, d in enumerate(data):
        if d['sphere_image'] is None:
            d['cnn_prediction'] = None                                       # :287-288
        _dump_pickle(d, files[i])
    print("finished ", dataset['destination_folder'])


def run_em_batch(data, distance_measure="angle", use_weights=True, do_split=True, do_merge=True, device=0,
                 defer_errors=False, store_distribution=False):
    """EM over a list of datum dicts in ONE launch; fills datum['EM_result'] like run_em_single.
    An image without any initial VP raises ValueError like the reference (vp_localisation.py:165) -- after
    the results of all other images have been filled in; with ``defer_errors`` the positions of such images
    are returned instead (their EM_result stays None).  ``store_distribution`` fills EM_result['distribution'] with the
    reference's PDF tuple (vp_localisation.py:441; three N x 64 fp64 arrays per image on the device while the batch
    runs, which is why whole-dataset runs leave it None -- run_em_single keeps it like the reference)."""
    todo = [d for d in data if d.get('cnn_prediction') is not None]
    scenes = [{"l": d['lines']['lines'], "lp": d['lines']['line_segments'],
               "cnn_response": d['cnn_prediction'][:, :], "sphere_image": d['sphere_image']} for d in todo]
    results = _em.em_batch(scenes, device=device, want_metric=True, want_distribution=store_distribution,
                           distance_measure=distance_measure, use_weights=use_weights, do_split=do_split,
                           do_merge=do_merge) if scenes else []
    failed = []
    for d, r in zip(todo, results):
        status = r.pop("status")
        d['lines']['lines'][...] = r.pop("l")            # the reference normalises l in place (:339,:350)
        if status == 2:
            d['EM_result'] = None
            failed.append(next(i for i, x in enumerate(data) if x is d))
            continue
        d['EM_result'] = r
    for d in data:
        if d.get('cnn_prediction') is None:
            d['EM_result'] = None                                            # :351-352
    if defer_errors:
        return failed
    if failed:
        raise ValueError("need at least one array to concatenate")          # vp_localisation.py:165
    return data


def run_em(dataset, start=None, end=None, indices=None, device=0, store_distribution=False):
    """evaluation.py:295-329.  ``start``/``end`` slice the file list like the reference (:304-307: its only
    means of spreading a dataset over several processes); ``indices`` selects an arbitrary subset instead
    (one rank's share of a cost-balanced partition, sharding.shard_balanced) and ``device`` the GPU."""
    files = [f[0] if isinstance(f, tuple) else f for f in dataset['pickle_files']]
    if indices is not None:
        files = [files[int(i)] for i in indices]
    elif not (start is None or end is None):
        files = files[start:min(end, len(files))]
    data = [_load_pickle(f) for f in files]
    failed = run_em_batch(data, distance_measure=dataset['distance_measure'], use_weights=dataset['use_weights'],
                          do_split=dataset['do_split'], do_merge=dataset['do_merge'], device=device, defer_errors=True,
                          store_distribution=store_distribution)
    for f, d in zip(files, data):
        if d.get('EM_result') is None:
            print("SKIPPING: file %s is incomplete" % f)
        _dump_pickle(d, f)
    if failed:
        # the reference dies with this ValueError at the first such image (vp_localisation.py:165) after
        # having stored every earlier one; here every other image of the dataset is stored first
        raise ValueError("need at least one array to concatenate (no initial VP in: %s)"
                         % ", ".join(files[i] for i in failed))


def run_em_single(datum, distance_measure="angle", use_weights=True, do_split=True, do_merge=True):
    """evaluation.py:332-354."""
    lines = datum['lines']
    if datum['cnn_prediction'] is not None:
        datum['EM_result'] = vp.expectation_maximisation(
            lines['lines'], lines['line_segments'], datum['cnn_prediction'][:, :],
            sphere_image=datum['sphere_image'], distance_measure=distance_measure, use_weights=use_weights,
            do_split=do_split, do_merge=do_merge)
        datum['lines'] = lines
    else:
        datum['EM_result'] = None
    return datum


def renew_cnn_result(net, mean_arr, lines, image_size):
    """evaluation.py:357-361."""
    image = get_sphere_image(lines, size=image_size)
    return (image, caffe_forward(net, image, mean_arr))
