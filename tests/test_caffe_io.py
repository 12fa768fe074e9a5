"""Caffe model I/O without Caffe: wire-format round trip and the deploy.prototxt topology check."""
import os

import numpy as np
import pytest

from vanishing_points_2017_amd import caffe_io

DEPLOY = """name: "AlexNet_for_VP_classification"
layer { name: "data" type: "Input" top: "data"
  input_param { shape: { dim: 1 dim: 1 dim: 500 dim: 500 } }
}
%s
"""


def _conv(name, n, k, s=1, p=0, g=1):
    extra = ("    stride: %d\n" % s if s != 1 else "") + ("    pad: %d\n" % p if p else "") + ("    group: %d\n" % g if g != 1 else "")
    return 'layer {\n  name: "%s"\n  type: "Convolution"\n  convolution_param {\n    num_output: %d\n    kernel_size: %d\n%s  }\n}\n' % (name, n, k, extra)


def _fc(name, n):
    return 'layer {\n  name: "%s"\n  type: "InnerProduct"\n  inner_product_param {\n    num_output: %d\n  }\n}\n' % (name, n)


def _deploy_text(conv2_group=2):
    body = _conv("conv1", 96, 11, 4) + _conv("conv2", 256, 5, 1, 2, conv2_group) + _conv("conv3", 384, 3, 1, 1) + \
        _conv("conv4", 384, 3, 1, 1, 2) + _conv("conv5", 256, 3, 1, 1, 2) + _fc("fc6", 4096) + _fc("fc7", 4096) + _fc("fc8_20x20", 400)
    return DEPLOY % body


def test_roundtrip(tmp_path):
    rs = np.random.RandomState(0)
    layers = {"conv1": [rs.randn(96, 1, 11, 11).astype(np.float32), rs.randn(96).astype(np.float32)],
              "fc8_20x20": [rs.randn(400, 64).astype(np.float32), rs.randn(400).astype(np.float32)]}
    path = str(tmp_path / "w.caffemodel")
    caffe_io.write_caffemodel(path, layers)
    back = caffe_io.read_caffemodel(path)
    assert set(back) == set(layers)
    for k in layers:
        for a, b in zip(layers[k], back[k]):
            assert a.shape == b.shape and np.array_equal(a, b)
    mean = rs.rand(1, 1, 500, 500).astype(np.float32)
    mpath = str(tmp_path / "mean.binaryproto")
    caffe_io.write_binaryproto(mpath, mean)
    got = caffe_io.read_binaryproto(mpath)
    assert got.shape == (1, 1, 500, 500) and np.array_equal(got, mean)


def test_prototxt_check(tmp_path):
    good = tmp_path / "deploy.prototxt"
    good.write_text(_deploy_text())
    assert caffe_io.check_deploy_prototxt(str(good))
    bad = tmp_path / "bad.prototxt"
    bad.write_text(_deploy_text(conv2_group=1))
    with pytest.raises(ValueError):
        caffe_io.check_deploy_prototxt(str(bad))


def test_reference_deploy_prototxt_if_present():
    path = "/root/reference/cnn/deploy.prototxt"          # build container only
    if not os.path.isfile(path):
        pytest.skip("reference tree not present (GPU box)")
    assert caffe_io.check_deploy_prototxt(path)


def test_reference_result_pickles_load(tmp_path):
    """A datum written by the reference holds EM_result['distribution'] = probability_functions.PDF(...)
    (vp_localisation.py:441) pickled under the top-level module name; evaluation._load_pickle resolves it."""
    import pickle
    import sys
    import types
    from collections import namedtuple
    import numpy as np
    from vanishing_points_2017_amd import evaluation
    fake = types.ModuleType("probability_functions")          # stands in for the reference's module while writing
    fake.PDF = namedtuple("PDF", "v lv vl l lvsq angles")
    fake.PDF.__module__ = "probability_functions"
    sys.modules["probability_functions"] = fake
    try:
        datum = {"EM_result": {"vp": np.eye(3), "distribution": fake.PDF(np.ones(3), None, None, None, None, None)},
                 "cnn_prediction": np.zeros((20, 20), np.float32)}
        path = tmp_path / "datum.pkl"
        with open(path, "wb") as fp:
            pickle.dump(datum, fp, 2)
    finally:
        del sys.modules["probability_functions"]
    got = evaluation._load_pickle(str(path))
    assert type(got["EM_result"]["distribution"]).__name__ == "PDF"
    assert got["EM_result"]["distribution"]._fields == ("v", "lv", "vl", "l", "lvsq", "angles")
    assert np.array_equal(got["EM_result"]["vp"], np.eye(3))
