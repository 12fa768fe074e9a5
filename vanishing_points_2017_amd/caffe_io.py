"""Minimal Caffe model I/O without Caffe: weights.caffemodel (NetParameter), mean.binaryproto
(BlobProto) and a topology check of deploy.prototxt (reference call sites
evaluation.py:17-31, config.py:6-8).

Only the protobuf *wire format* is parsed (varint / 64-bit / length-delimited / 32-bit).  Field
numbers are those of upstream BVLC caffe.proto (they are not in the reference repository):
  NetParameter:      name=1, layers=2 (V1LayerParameter), layer=100 (LayerParameter)
  LayerParameter:    name=1, type=2, blobs=7
  V1LayerParameter:  name=4, blobs=6
  BlobProto:         num=1, channels=2, height=3, width=4, data=5 (packed float), shape=7,
                     double_data=8 (packed double)
  BlobShape:         dim=1 (packed int64)
A writer for the same subset exists so tests can round-trip files without Caffe; the reader is also pinned against files
encoded by Google's protobuf runtime from upstream's field numbers (tests/golden/caffe_proto.npz, written by a
build-container script of the test infrastructure: `layer` and V1 `layers` messages, packed and unpacked arrays, legacy 4-D dims,
double_data, BlobShape).
"""
import re
import struct

import numpy as np


# ---- wire format ---------------------------------------------------------------------------------
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one message; value is int or memoryview."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fnum, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fnum, wt, val


def _parse_blob(buf):
    dims_legacy = {}
    shape = None
    chunks = []
    dchunks = []
    for fnum, wt, val in _fields(buf):
        if fnum in (1, 2, 3, 4) and wt == 0:
            dims_legacy[fnum] = val
        elif fnum == 5:
            chunks.append(np.frombuffer(val, dtype="<f4") if wt == 2 else np.frombuffer(val, dtype="<f4", count=1))
        elif fnum == 8:
            dchunks.append(np.frombuffer(val, dtype="<f8") if wt == 2 else np.frombuffer(val, dtype="<f8", count=1))
        elif fnum == 7 and wt == 2:
            dims = []
            for f2, w2, v2 in _fields(val):
                if f2 == 1 and w2 == 2:
                    p = 0
                    while p < len(v2):
                        d, p = _varint(v2, p)
                        dims.append(d)
                elif f2 == 1 and w2 == 0:
                    dims.append(v2)
            shape = tuple(dims)
    if chunks:
        data = np.concatenate(chunks).astype(np.float32)
    elif dchunks:
        data = np.concatenate(dchunks).astype(np.float32)
    else:
        data = np.zeros(0, np.float32)
    if shape is None:
        shape = tuple(dims_legacy.get(k, 1) for k in (1, 2, 3, 4)) if dims_legacy else (data.size,)
    return data.reshape(shape) if int(np.prod(shape)) == data.size else data


def read_caffemodel(path):
    """{layer name: [blob arrays]} for every layer that carries blobs."""
    return parse_caffemodel(open(path, "rb").read())


def parse_caffemodel(data):
    """read_caffemodel on the file's bytes (a serialised NetParameter)."""
    buf = memoryview(data)
    layers = {}
    for fnum, wt, val in _fields(buf):
        if wt != 2 or fnum not in (2, 100):
            continue
        name_field, blob_field = (1, 7) if fnum == 100 else (4, 6)
        name = None
        blobs = []
        for f2, w2, v2 in _fields(val):
            if f2 == name_field and w2 == 2:
                name = bytes(v2).decode("utf-8", "replace")
            elif f2 == blob_field and w2 == 2:
                blobs.append(_parse_blob(v2))
        if name is not None and blobs:
            layers[name] = blobs
    return layers


def read_binaryproto(path):
    """mean.binaryproto -> ndarray shaped like caffe.io.blobproto_to_array (num, channels, h, w)."""
    return parse_binaryproto(open(path, "rb").read())


def parse_binaryproto(data):
    """read_binaryproto on the file's bytes (a serialised BlobProto)."""
    arr = _parse_blob(memoryview(data))
    if arr.ndim == 3:
        arr = arr[None]
    return arr


# ---- writer (tests / tooling) ----------------------------------------------------------------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(fnum, payload):
    return _enc_varint((fnum << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_blob(arr):
    arr = np.ascontiguousarray(arr, dtype="<f4")
    shape = _enc_ld(1, b"".join(_enc_varint(int(d)) for d in arr.shape))
    return _enc_ld(5, arr.tobytes()) + _enc_ld(7, shape)      # field-number order, as protobuf serialises it


def write_binaryproto(path, arr):
    with open(path, "wb") as fh:
        fh.write(_enc_blob(arr))


def write_caffemodel(path, layers, name="net"):
    """layers: ordered {layer name: [arrays]} -> NetParameter with LayerParameter entries (field 100)."""
    with open(path, "wb") as fh:
        fh.write(_enc_ld(1, name.encode()))
        for lname, blobs in layers.items():
            body = _enc_ld(1, lname.encode()) + _enc_ld(2, b"Layer")
            for b in blobs:
                body += _enc_ld(7, _enc_blob(b))
            fh.write(_enc_ld(100, body))


# ---- deploy.prototxt topology check ------------------------------------------------------------------
EXPECTED_TOPOLOGY = {   # cnn/deploy.prototxt: name -> (num_output, kernel, stride, pad, group)
    "conv1": (96, 11, 4, 0, 1), "conv2": (256, 5, 1, 2, 2), "conv3": (384, 3, 1, 1, 1),
    "conv4": (384, 3, 1, 1, 2), "conv5": (256, 3, 1, 1, 2),
    "fc6": (4096,), "fc7": (4096,), "fc8_20x20": (400,),
}


def check_deploy_prototxt(path):
    """Raise ValueError unless the prototxt describes the one topology the kernels implement."""
    text = open(path).read()
    blocks = re.findall(r"layer\s*\{(.*?)\n\}", text, flags=re.S)
    found = {}
    for blk in blocks:
        m = re.search(r'name:\s*"([^"]+)"', blk)
        if not m:
            continue
        name = m.group(1)
        num = lambda key, default: int((re.search(key + r":\s*(\d+)", blk) or [None, default])[1])
        if "convolution_param" in blk:
            found[name] = (num("num_output", 0), num("kernel_size", 0), num("stride", 1), num("pad", 0), num("group", 1))
        elif "inner_product_param" in blk:
            found[name] = (num("num_output", 0),)
    for name, want in EXPECTED_TOPOLOGY.items():
        if found.get(name) != want:
            raise ValueError("deploy.prototxt layer %s is %s, the HIP kernels implement %s" % (name, found.get(name), want))
    dims = re.search(r"dim:\s*1\s+dim:\s*1\s+dim:\s*500\s+dim:\s*500", text)
    if not dims:
        raise ValueError("deploy.prototxt input is not 1x1x500x500")
    return True
