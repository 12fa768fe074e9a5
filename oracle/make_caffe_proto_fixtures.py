#!/usr/bin/env python3
"""Third-party-encoded Caffe model files for tests/test_caffe_io.py (test infrastructure; build container only).

vanishing_points_2017_amd/caffe_io.py reads weights.caffemodel / mean.binaryproto (the reference's call sites:
evaluation.py:17-31, config.py:6-8) by parsing the protobuf wire format itself, and until now was only checked against
its own writer.  This script encodes small nets with GOOGLE'S protobuf runtime (`google.protobuf`, importable here) from a
descriptor built below out of upstream BVLC caffe.proto's message and field numbers (they are not in the reference
repository; Caffe 1.0's src/caffe/proto/caffe.proto):

    BlobShape        dim = 1 (repeated int64, packed)
    BlobProto        num = 1, channels = 2, height = 3, width = 4 (int32), data = 5 / diff = 6 (repeated float, packed),
                     shape = 7 (BlobShape), double_data = 8 / double_diff = 9 (repeated double, packed)
    ParamSpec        name = 1, lr_mult = 3 (float), decay_mult = 4 (float)
    ConvolutionParameter  num_output = 1, pad = 3, kernel_size = 4, group = 5, stride = 6 (uint32; pad / kernel / stride repeated)
    LayerParameter   name = 1, type = 2, bottom = 3, top = 4, loss_weight = 5, param = 6 (ParamSpec), blobs = 7 (BlobProto),
                     phase = 10 (enum), convolution_param = 106
    V1LayerParameter bottom = 2, top = 3, name = 4, type = 5 (enum LayerType, CONVOLUTION = 4, INNER_PRODUCT = 14), blobs = 6,
                     blobs_lr = 7, weight_decay = 8
    NetParameter     name = 1, layers = 2 (V1LayerParameter), input = 3, input_dim = 4, layer = 100 (LayerParameter)

and stores the encoded BYTES with the arrays they hold in tests/golden/caffe_proto.npz.  Cases: `layer` (100) messages with
BlobShape + packed data, legacy num / channels / height / width, double_data, a diff beside the data, fields the reader
must skip (strings, varints, 32-bit floats, nested messages); V1 `layers` (2) messages; UNPACKED repeated floats (a
second descriptor whose data / double_data are declared without [packed = true]: what an old writer emits and every
reader must accept); mean blobs in the legacy 4-D and in the BlobShape form.

    python oracle/make_caffe_proto_fixtures.py        # writes tests/golden/caffe_proto.npz
"""
import os

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, packed=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    return f


def build_messages(package, packed):
    """caffe.proto's subset as a FileDescriptorProto in `package`; packed: the [packed = ...] option of the blob arrays."""
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = package + ".proto"
    fd.package = package
    fd.syntax = "proto2"
    pre = "." + package + "."
    rep = F.LABEL_REPEATED

    m = fd.message_type.add(); m.name = "BlobShape"
    _field(m, "dim", 1, F.TYPE_INT64, rep, packed=True)

    m = fd.message_type.add(); m.name = "BlobProto"
    _field(m, "num", 1, F.TYPE_INT32)
    _field(m, "channels", 2, F.TYPE_INT32)
    _field(m, "height", 3, F.TYPE_INT32)
    _field(m, "width", 4, F.TYPE_INT32)
    _field(m, "data", 5, F.TYPE_FLOAT, rep, packed=packed)
    _field(m, "diff", 6, F.TYPE_FLOAT, rep, packed=packed)
    _field(m, "shape", 7, F.TYPE_MESSAGE, type_name=pre + "BlobShape")
    _field(m, "double_data", 8, F.TYPE_DOUBLE, rep, packed=packed)
    _field(m, "double_diff", 9, F.TYPE_DOUBLE, rep, packed=packed)

    m = fd.message_type.add(); m.name = "ParamSpec"
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "lr_mult", 3, F.TYPE_FLOAT)
    _field(m, "decay_mult", 4, F.TYPE_FLOAT)

    m = fd.message_type.add(); m.name = "ConvolutionParameter"
    _field(m, "num_output", 1, F.TYPE_UINT32)
    _field(m, "pad", 3, F.TYPE_UINT32, rep)
    _field(m, "kernel_size", 4, F.TYPE_UINT32, rep)
    _field(m, "group", 5, F.TYPE_UINT32)
    _field(m, "stride", 6, F.TYPE_UINT32, rep)

    e = fd.enum_type.add(); e.name = "Phase"
    for n, v in (("TRAIN", 0), ("TEST", 1)):
        x = e.value.add(); x.name, x.number = n, v

    m = fd.message_type.add(); m.name = "LayerParameter"
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "type", 2, F.TYPE_STRING)
    _field(m, "bottom", 3, F.TYPE_STRING, rep)
    _field(m, "top", 4, F.TYPE_STRING, rep)
    _field(m, "loss_weight", 5, F.TYPE_FLOAT, rep)
    _field(m, "param", 6, F.TYPE_MESSAGE, rep, type_name=pre + "ParamSpec")
    _field(m, "blobs", 7, F.TYPE_MESSAGE, rep, type_name=pre + "BlobProto")
    _field(m, "phase", 10, F.TYPE_ENUM, type_name=pre + "Phase")
    _field(m, "convolution_param", 106, F.TYPE_MESSAGE, type_name=pre + "ConvolutionParameter")

    m = fd.message_type.add(); m.name = "V1LayerParameter"
    e = m.enum_type.add(); e.name = "LayerType"
    for n, v in (("NONE", 0), ("CONVOLUTION", 4), ("INNER_PRODUCT", 14), ("RELU", 18)):
        x = e.value.add(); x.name, x.number = n, v
    _field(m, "bottom", 2, F.TYPE_STRING, rep)
    _field(m, "top", 3, F.TYPE_STRING, rep)
    _field(m, "name", 4, F.TYPE_STRING)
    _field(m, "type", 5, F.TYPE_ENUM, type_name=pre + "V1LayerParameter.LayerType")
    _field(m, "blobs", 6, F.TYPE_MESSAGE, rep, type_name=pre + "BlobProto")
    _field(m, "blobs_lr", 7, F.TYPE_FLOAT, rep)
    _field(m, "weight_decay", 8, F.TYPE_FLOAT, rep)

    m = fd.message_type.add(); m.name = "NetParameter"
    _field(m, "name", 1, F.TYPE_STRING)
    _field(m, "layers", 2, F.TYPE_MESSAGE, rep, type_name=pre + "V1LayerParameter")
    _field(m, "input", 3, F.TYPE_STRING, rep)
    _field(m, "input_dim", 4, F.TYPE_INT32, rep)
    _field(m, "layer", 100, F.TYPE_MESSAGE, rep, type_name=pre + "LayerParameter")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda name: message_factory.GetMessageClass(pool.FindMessageTypeByName(package + "." + name))
    return {n: get(n) for n in ("BlobShape", "BlobProto", "LayerParameter", "V1LayerParameter", "NetParameter")}


def fill_blob(bp, arr, form):
    """form: 'shape' (BlobShape + data), 'legacy' (num/channels/height/width + data), 'double' (BlobShape + double_data),
    'shape+diff' (as 'shape' with a diff of the same size that a reader must not mix up with the data)"""
    arr = np.asarray(arr)
    if form == "legacy":
        d = (1,) * (4 - arr.ndim) + arr.shape
        bp.num, bp.channels, bp.height, bp.width = [int(x) for x in d]
    else:
        bp.shape.dim.extend(int(x) for x in arr.shape)
    if form == "double":
        bp.double_data.extend(float(x) for x in arr.reshape(-1).astype(np.float64))
    else:
        bp.data.extend(float(x) for x in arr.reshape(-1).astype(np.float32))
    if form == "shape+diff":
        bp.diff.extend(float(x) for x in (arr.reshape(-1) * -3.0 + 7.0).astype(np.float32))


def main():
    rs = np.random.RandomState(20171707)
    out = {}
    arrays = {   # what the files hold (float32, as the reader returns them)
        "conv1_w": rs.randn(4, 1, 3, 3).astype(np.float32), "conv1_b": rs.randn(4).astype(np.float32),
        "conv2_w": rs.randn(6, 2, 5, 5).astype(np.float32), "conv2_b": rs.randn(6).astype(np.float32),
        "fc6_w": rs.randn(5, 36).astype(np.float32), "fc6_b": rs.randn(5).astype(np.float32),
        "fc8_w": rs.randn(3, 5).astype(np.float32), "fc8_b": rs.randn(3).astype(np.float32),
        "mean": (rs.rand(1, 1, 6, 7) * 255).astype(np.float32),
    }
    for k, v in arrays.items():
        out["want_" + k] = v

    for tag, packed in (("packed", True), ("unpacked", False)):
        M = build_messages("caffe_" + tag, packed)
        # ---- NetParameter with `layer` (100) messages ----
        net = M["NetParameter"]()
        net.name = "tiny_" + tag
        net.input.append("data")
        net.input_dim.extend([1, 1, 12, 12])
        lay = net.layer.add()
        lay.name, lay.type = "data", "Input"
        lay.top.append("data")
        lay = net.layer.add()
        lay.name, lay.type = "conv1", "Convolution"
        lay.bottom.append("data"); lay.top.append("conv1")
        lay.phase = 1
        lay.loss_weight.append(0.5)
        for lr, dec in ((1.0, 1.0), (2.0, 0.0)):
            ps = lay.param.add(); ps.lr_mult, ps.decay_mult = lr, dec
        lay.convolution_param.num_output = 4
        lay.convolution_param.kernel_size.append(3)
        lay.convolution_param.stride.append(1)
        fill_blob(lay.blobs.add(), arrays["conv1_w"], "shape+diff")
        fill_blob(lay.blobs.add(), arrays["conv1_b"], "shape")
        lay = net.layer.add()
        lay.name, lay.type = "relu1", "ReLU"          # a layer without blobs
        lay.bottom.append("conv1"); lay.top.append("conv1")
        lay = net.layer.add()
        lay.name, lay.type = "conv2", "Convolution"
        lay.convolution_param.num_output, lay.convolution_param.group = 6, 1
        fill_blob(lay.blobs.add(), arrays["conv2_w"], "double")
        fill_blob(lay.blobs.add(), arrays["conv2_b"], "double")
        lay = net.layer.add()
        lay.name, lay.type = "fc6", "InnerProduct"
        fill_blob(lay.blobs.add(), arrays["fc6_w"], "legacy")     # (1, 1, 5, 36), as pre-BlobShape Caffe wrote dense weights
        fill_blob(lay.blobs.add(), arrays["fc6_b"], "legacy")     # (1, 1, 1, 5)
        out["net_layer_" + tag] = np.frombuffer(net.SerializeToString(), dtype=np.uint8)
        # ---- NetParameter with V1 `layers` (2) messages ----
        v1 = M["NetParameter"]()
        v1.name = "tiny_v1_" + tag
        lay = v1.layers.add()
        lay.name, lay.type = "conv1", 4
        lay.bottom.append("data"); lay.top.append("conv1")
        lay.blobs_lr.extend([1.0, 2.0]); lay.weight_decay.extend([1.0, 0.0])
        fill_blob(lay.blobs.add(), arrays["conv1_w"], "legacy")
        fill_blob(lay.blobs.add(), arrays["conv1_b"], "legacy")
        lay = v1.layers.add()
        lay.name, lay.type = "relu1", 18
        lay = v1.layers.add()
        lay.name, lay.type = "fc8_20x20", 14
        fill_blob(lay.blobs.add(), arrays["fc8_w"], "legacy")
        fill_blob(lay.blobs.add(), arrays["fc8_b"], "shape")
        out["net_v1_" + tag] = np.frombuffer(v1.SerializeToString(), dtype=np.uint8)
        # ---- mean.binaryproto: a bare BlobProto ----
        for form in ("legacy", "shape", "double"):
            bp = M["BlobProto"]()
            fill_blob(bp, arrays["mean"], form)
            out["mean_%s_%s" % (form, tag)] = np.frombuffer(bp.SerializeToString(), dtype=np.uint8)

    import google.protobuf
    out["protobuf_version"] = np.frombuffer(google.protobuf.__version__.encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "caffe_proto.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.size for k, v in out.items() if not k.startswith("want_")})


if __name__ == "__main__":
    main()
