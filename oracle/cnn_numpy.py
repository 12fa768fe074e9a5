"""Second, independent CPU restatement of the CNN's layer semantics: float64 NumPy, direct loops.

TEST INFRASTRUCTURE (see oracle/em_numpy.py for the rules).  The first oracle, oracle/cnn_torch.py, maps
cnn/deploy.prototxt onto torch's functional ops and trusts that torch's conventions equal Caffe's.  The
reference runs the net inside third-party BVLC Caffe 1.0-RC5 (README.md:5; evaluation.py:17-38), which is
not installable here -- PARITY UNPINNED at that boundary.  This file restates each layer type from Caffe's
published definition with explicit index arithmetic, small enough to check by hand, so that the two oracles
can be checked against each other (tests/test_cnn_oracles.py) on the conventions a port can get wrong:

  Convolution (deploy.prototxt:9-27,56-75,104-174; Caffe base_conv_layer): cross-correlation (no kernel
      flip), weights (out, in/group, kh, kw), `group` splits input AND output channels into contiguous
      blocks, out = floor((H + 2 pad - k) / stride) + 1.
  Pooling MAX (:45-55,92-103,181-191; Caffe pooling_layer): out = CEIL((H + 2 pad - k) / stride) + 1; with
      pad > 0 the last window must start inside the padded input; each window is clipped to the input.
  LRN ACROSS_CHANNELS (:34-44,82-91; Caffe lrn_layer): scale_c = k + (alpha / n) * sum of x^2 over the n
      channels centred on c (missing channels count as 0), y = x * scale^-beta; k = 1.
  InnerProduct (:192-281; Caffe inner_product_layer): y = W x + b, W (out, in), x = the bottom blob
      flattened in (C, H, W) order.
  Dropout (:211-223,243-255) is the identity at TEST; Reshape (:283-296) a view; Sigmoid (:298-304).
"""
import numpy as np


def conv2d(x, w, b, stride=1, pad=0, group=1):
    """x (C, H, W), w (OC, C/group, KH, KW), b (OC) -> (OC, OH, OW), float64."""
    x = np.asarray(x, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    c_in, h, wd = x.shape
    oc, icg, kh, kw = w.shape
    assert c_in == icg * group and oc % group == 0
    ocg = oc // group
    oh = (h + 2 * pad - kh) // stride + 1
    ow = (wd + 2 * pad - kw) // stride + 1
    xp = np.zeros((c_in, h + 2 * pad, wd + 2 * pad))
    xp[:, pad:pad + h, pad:pad + wd] = x
    out = np.zeros((oc, oh, ow))
    for o in range(oc):
        g = o // ocg                                     # output channel o belongs to group g ...
        for y in range(oh):
            for xx in range(ow):
                acc = float(b[o])
                for c in range(icg):                     # ... and sees input channels g*icg .. (g+1)*icg - 1 only
                    patch = xp[g * icg + c, y * stride:y * stride + kh, xx * stride:xx * stride + kw]
                    acc += float((patch * w[o, c]).sum())    # cross-correlation: no flip
                out[o, y, xx] = acc
    return out


def relu(x):
    return np.maximum(x, 0.0)


def max_pool(x, k=3, stride=2, pad=0):
    """Caffe pooling_layer: ceil-mode output size, windows clipped to the input."""
    x = np.asarray(x, dtype=np.float64)
    c, h, w = x.shape

    def out_size(n):
        o = int(np.ceil((n + 2 * pad - k) / float(stride))) + 1
        if pad > 0 and (o - 1) * stride >= n + pad:      # last window must start inside the (padded) input
            o -= 1
        return o
    ph, pw = out_size(h), out_size(w)
    out = np.zeros((c, ph, pw))
    for i in range(ph):
        h0 = i * stride - pad
        h1 = min(h0 + k, h + pad)
        h0, h1 = max(h0, 0), min(h1, h)
        for j in range(pw):
            w0 = j * stride - pad
            w1 = min(w0 + k, w + pad)
            w0, w1 = max(w0, 0), min(w1, w)
            out[:, i, j] = x[:, h0:h1, w0:w1].reshape(c, -1).max(axis=1)
    return out


def lrn_across_channels(x, n=5, alpha=1e-4, beta=0.75, k=1.0):
    x = np.asarray(x, dtype=np.float64)
    c = x.shape[0]
    out = np.zeros_like(x)
    half = (n - 1) // 2
    for ch in range(c):
        lo, hi = max(0, ch - half), min(c - 1, ch + half)
        scale = k + (alpha / n) * (x[lo:hi + 1] ** 2).sum(axis=0)     # always divided by n, also at the edges
        out[ch] = x[ch] * scale ** (-beta)
    return out


def inner_product(x, w, b):
    """x: any (C, H, W) blob, flattened in that order."""
    return np.asarray(w, dtype=np.float64).dot(np.asarray(x, dtype=np.float64).reshape(-1)) + np.asarray(b, dtype=np.float64)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64)))


def forward_small(weights, x):
    """The deploy.prototxt layer sequence on an arbitrary small input blob x (1, H, W) -- already mean-subtracted
    (evaluation.py:35) -- down to pool5; the fully connected layers need the full 500 x 500 geometry and are
    covered by inner_product on their own.  weights: {name: (W, b)} (any channel counts that chain)."""
    t = {}
    a = relu(conv2d(x, *weights["conv1"], stride=4)); t["conv1"] = a
    a = max_pool(lrn_across_channels(a)); t["pool1"] = a
    a = relu(conv2d(a, *weights["conv2"], pad=2, group=2)); t["conv2"] = a
    a = max_pool(lrn_across_channels(a)); t["pool2"] = a
    a = relu(conv2d(a, *weights["conv3"], pad=1)); t["conv3"] = a
    a = relu(conv2d(a, *weights["conv4"], pad=1, group=2)); t["conv4"] = a
    a = relu(conv2d(a, *weights["conv5"], pad=1, group=2)); t["conv5"] = a
    a = max_pool(a); t["pool5"] = a
    return t
