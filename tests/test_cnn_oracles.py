"""The two CPU restatements of the CNN (oracle/cnn_torch.py: torch functional ops; oracle/cnn_numpy.py:
float64 direct loops written from Caffe's layer definitions) against each other and against hand-computed
micro cases.  The Caffe boundary itself stays unpinned (Caffe and the trained weights are absent); what
this pins is that the torch oracle -- the witness of every HIP CNN test -- implements the conventions of
cnn/deploy.prototxt the way Caffe defines them: ceil-mode clipped pooling windows (:45-55), LRN at the
channel edges (:34-44), contiguous group split (:56-75), (out, in) weights over C*H*W (:192-210)."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cnn_numpy as cn


def test_pool_ceil_mode_by_hand():
    # 4 x 4 input, 3/2 pooling: Caffe gives ceil((4-3)/2)+1 = 2 outputs per axis; the second window covers
    # rows/cols 2..3 only (clipped) -- a floor-mode port would return 1 x 1
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4)
    got = cn.max_pool(x)
    assert got.shape == (1, 2, 2)
    assert np.array_equal(got[0], [[10.0, 11.0], [14.0, 15.0]])
    # the net's own sizes: 123 -> 61, 61 -> 30, 30 -> 15 (deploy.prototxt pools)
    for n, want in ((123, 61), (61, 30), (30, 15)):
        assert cn.max_pool(np.zeros((1, n, n))).shape == (1, want, want)
    # 61 -> 30 has NO partial window ((61-3)/2 is integral); 123 -> 61 likewise; 30 -> 15 has one (rows 28..29)
    x = np.zeros((1, 30, 30)); x[0, 29, 29] = 7.0
    assert cn.max_pool(x)[0, 14, 14] == 7.0 and cn.max_pool(x)[0, 13, 13] == 0.0


def test_lrn_edges_by_hand():
    # 3 channels, n = 5: every window is clipped, yet the divisor stays n = 5 (Caffe: alpha / local_size)
    x = np.array([1.0, 2.0, 3.0]).reshape(3, 1, 1)
    got = cn.lrn_across_channels(x, n=5, alpha=0.5, beta=0.75, k=1.0).reshape(3)
    s = 1.0 + 0.1 * (1 + 4 + 9)
    assert np.allclose(got, np.array([1.0, 2.0, 3.0]) * s ** -0.75, rtol=1e-15)
    # 7 channels: channel 0 sees 0..2, channel 3 sees 1..5
    x = np.arange(1.0, 8.0).reshape(7, 1, 1)
    got = cn.lrn_across_channels(x, n=5, alpha=1.0, beta=1.0, k=1.0).reshape(7)
    assert np.isclose(got[0], 1.0 / (1 + (1 + 4 + 9) / 5.0))
    assert np.isclose(got[3], 4.0 / (1 + (4 + 9 + 16 + 25 + 36) / 5.0))


def test_group_split_by_hand():
    # group = 2, 4 inputs -> 2 outputs: output 0 may only see inputs 0,1; output 1 only inputs 2,3
    x = np.zeros((4, 1, 1)); x[:, 0, 0] = [1, 10, 100, 1000]
    w = np.ones((2, 2, 1, 1))
    got = cn.conv2d(x, w, np.zeros(2), group=2).reshape(2)
    assert np.array_equal(got, [11.0, 1100.0])
    # cross-correlation: the kernel is NOT flipped
    x = np.zeros((1, 2, 2)); x[0] = [[1, 2], [3, 4]]
    w = np.zeros((1, 1, 2, 2)); w[0, 0] = [[1, 0], [0, 0]]
    assert cn.conv2d(x, w, np.zeros(1))[0, 0, 0] == 1.0


def test_inner_product_flatten_order():
    x = np.arange(2 * 3 * 4, dtype=np.float64).reshape(2, 3, 4)
    w = np.zeros((1, 24)); w[0, 1 * 12 + 2 * 4 + 3] = 1.0          # picks element (c=1, h=2, w=3)
    assert cn.inner_product(x, w, np.zeros(1))[0] == x[1, 2, 3]


def _torch_layers(weights, x):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
    w = {k: (t(v[0]), t(v[1])) for k, v in weights.items()}
    a = t(x)[None]
    out = {}
    with torch.no_grad():
        a = F.relu(F.conv2d(a, *w["conv1"], stride=4)); out["conv1"] = a
        a = F.max_pool2d(F.local_response_norm(a, 5, alpha=1e-4, beta=0.75, k=1.0), 3, 2, ceil_mode=True); out["pool1"] = a
        a = F.relu(F.conv2d(a, *w["conv2"], padding=2, groups=2)); out["conv2"] = a
        a = F.max_pool2d(F.local_response_norm(a, 5, alpha=1e-4, beta=0.75, k=1.0), 3, 2, ceil_mode=True); out["pool2"] = a
        a = F.relu(F.conv2d(a, *w["conv3"], padding=1)); out["conv3"] = a
        a = F.relu(F.conv2d(a, *w["conv4"], padding=1, groups=2)); out["conv4"] = a
        a = F.relu(F.conv2d(a, *w["conv5"], padding=1, groups=2)); out["conv5"] = a
        a = F.max_pool2d(a, 3, 2, ceil_mode=True); out["pool5"] = a
    return {k: v[0].numpy() for k, v in out.items()}


def test_torch_oracle_ops_equal_the_direct_loops_on_a_small_net():
    """deploy.prototxt's layer sequence with its kernel sizes / strides / pads / groups / LRN constants on a
    79 x 79 crop and thinned channel counts (12 / 16 / 24 / 24 / 16): torch's ops == the direct loops to 1e-12.
    79 -> conv1 18 -> pool1 9 (last window clipped) -> pool2 4 (exact fit) -> pool5 2 (clipped): every
    ceil-mode case occurs."""
    rs = np.random.RandomState(0)
    shapes = {"conv1": (12, 1, 11, 11), "conv2": (16, 6, 5, 5), "conv3": (24, 16, 3, 3), "conv4": (24, 12, 3, 3),
              "conv5": (16, 12, 3, 3)}
    weights = {k: (rs.standard_normal(s) * np.sqrt(2.0 / np.prod(s[1:])), rs.standard_normal(s[0]) * 0.1)
               for k, s in shapes.items()}
    x = rs.uniform(-20, 200, (1, 79, 79))
    weights["conv1"] = (weights["conv1"][0] / 40.0, weights["conv1"][1])
    a = cn.forward_small(weights, x)
    b = _torch_layers(weights, x)
    assert a["conv1"].shape == (12, 18, 18) and a["pool1"].shape == (12, 9, 9) and a["pool2"].shape == (16, 4, 4)
    assert a["pool5"].shape == (16, 2, 2)
    for k in a:
        assert a[k].shape == b[k].shape, k
        assert np.abs(a[k] - b[k]).max() <= 1e-12 * (1 + np.abs(a[k]).max()), k
        assert np.abs(a[k]).max() > 1e-3, k             # the comparison is not vacuous


def test_torch_oracle_module_matches_direct_loops_on_fc_and_sigmoid():
    rs = np.random.RandomState(1)
    x = rs.standard_normal((4, 3, 3))
    w, b = rs.standard_normal((5, 36)), rs.standard_normal(5)
    want = cn.sigmoid(cn.inner_product(x, w, b))
    got = torch.sigmoid(F.linear(torch.from_numpy(x).flatten()[None], torch.from_numpy(w), torch.from_numpy(b)))[0].numpy()
    assert np.abs(got - want).max() <= 1e-14
