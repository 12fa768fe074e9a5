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


# ---- the reader against files encoded by Google's protobuf runtime (oracle/make_caffe_proto_fixtures.py) ------------------
def _proto_fixtures():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caffe_proto.npz")
    return np.load(path, allow_pickle=False)


@pytest.mark.parametrize("tag", ["packed", "unpacked"])
def test_reader_on_protobuf_encoded_layer_messages(tag):
    """NetParameter.layer (field 100) as google.protobuf serialises it: BlobShape + data with a diff beside it, double_data,
    legacy num / channels / height / width, layers without blobs, and the fields a reader has to skip (strings, enums,
    32-bit floats, nested ParamSpec / ConvolutionParameter messages, repeated input / input_dim of the net)."""
    g = _proto_fixtures()
    layers = caffe_io.parse_caffemodel(g["net_layer_" + tag].tobytes())
    assert sorted(layers) == ["conv1", "conv2", "fc6"]                     # "data" and "relu1" carry no blobs
    for name, shapes in (("conv1", None), ("conv2", None), ("fc6", ((1, 1, 5, 36), (1, 1, 1, 5)))):
        w, b = layers[name]
        assert w.dtype == np.float32 and b.dtype == np.float32
        want_w, want_b = g["want_%s_w" % name], g["want_%s_b" % name]
        if shapes:                                                          # legacy 4-D blobs keep Caffe's leading ones
            assert (w.shape, b.shape) == shapes
        else:
            assert w.shape == want_w.shape and b.shape == want_b.shape
        assert np.array_equal(w.reshape(-1), want_w.reshape(-1)) and np.array_equal(b.reshape(-1), want_b.reshape(-1))


@pytest.mark.parametrize("tag", ["packed", "unpacked"])
def test_reader_on_protobuf_encoded_v1_layers(tag):
    """NetParameter.layers (field 2, V1LayerParameter: name = 4, blobs = 6) -- the format of models saved before Caffe's
    LayerParameter rewrite; evaluation.init_caffe loads whichever the file holds."""
    g = _proto_fixtures()
    layers = caffe_io.parse_caffemodel(g["net_v1_" + tag].tobytes())
    assert sorted(layers) == ["conv1", "fc8_20x20"]
    assert layers["conv1"][0].shape == (4, 1, 3, 3) and np.array_equal(layers["conv1"][0], g["want_conv1_w"])
    assert layers["conv1"][1].shape == (1, 1, 1, 4) and np.array_equal(layers["conv1"][1].reshape(-1), g["want_conv1_b"])
    assert layers["fc8_20x20"][0].shape == (1, 1, 3, 5) and np.array_equal(layers["fc8_20x20"][0].reshape(3, 5), g["want_fc8_w"])
    assert layers["fc8_20x20"][1].shape == (3,) and np.array_equal(layers["fc8_20x20"][1], g["want_fc8_b"])


@pytest.mark.parametrize("form", ["legacy", "shape", "double"])
@pytest.mark.parametrize("tag", ["packed", "unpacked"])
def test_reader_on_protobuf_encoded_mean_blobs(form, tag):
    """mean.binaryproto = one BlobProto (evaluation.py:25-31 turns it into a (1, 1, H, W) array)."""
    g = _proto_fixtures()
    got = caffe_io.parse_binaryproto(g["mean_%s_%s" % (form, tag)].tobytes())
    assert got.shape == (1, 1, 6, 7) and got.dtype == np.float32 and np.array_equal(got, g["want_mean"])


def test_own_writer_matches_the_protobuf_encoding():
    """caffe_io's writer (BlobShape + packed data) must produce what google.protobuf produces for the same blob: the two
    encodings of the mean are compared byte for byte."""
    g = _proto_fixtures()
    assert caffe_io._enc_blob(g["want_mean"]) == g["mean_shape_packed"].tobytes()
