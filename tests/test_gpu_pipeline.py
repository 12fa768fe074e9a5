"""End-to-end parity of the metric: horizon-line AUC of the GPU path vs the CPU oracle on the same
seeded synthetic YUD-shape inputs (BASELINE.json: 'identical horizon-line AUC', errors within 1e-4),
through the reference's call surface (evaluation.run_em on reference-schema pickles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_horizon_auc_parity_yud_shape(tmp_path):
    from oracle import em_numpy
    from vanishing_points_2017_amd import auc, benchmark, calc_horizon as ch, evaluation
    ds = benchmark.synthetic_dataset("york", str(tmp_path), 40, True)
    evaluation.run_em(ds)
    auc_gpu, err_gpu, _ = benchmark.score(ds, start=25)
    err_ref = []
    for idx, f in enumerate(ds['pickle_files']):
        if idx < 25:
            continue
        d = evaluation._load_pickle(f)
        lines = d['lines']
        # the pickled lines were normalised in place by run_em (like the reference): equivalent input
        ref = em_numpy.expectation_maximisation(lines['lines'].copy(), lines['line_segments'].copy(),
                                                d['cnn_prediction'].copy(), sphere_image=d['sphere_image'])
        got = d['EM_result']
        assert np.array_equal(got['vp_assoc'], ref['vp_assoc'])              # bit-exact assignments
        assert np.abs(got['vp'] - ref['vp']).max() <= 1e-4
        hp1, hp2, _, _, _, _ = ch.calculate_horizon_and_ortho_vp(ref, maxbest=20, theta_vmin=np.pi / 10)
        err_ref.append(ch.horizon_error(hp1, hp2, ds['true_horizon'][idx], ds['image_shape'][idx]))
    err_ref = np.array(err_ref)
    assert np.abs(err_gpu - err_ref).max() <= 1e-4
    auc_ref, _ = auc.calc_auc(err_ref, cutoff=0.25)
    assert abs(auc_gpu - auc_ref) <= 1e-4
    assert auc_gpu > 0.5                                                     # the scenes are solvable


def test_run_cnn_surface_with_written_model_files(tmp_path):
    """init_caffe / read_mean_blob / run_cnn on caffemodel + binaryproto files written without Caffe."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import benchmark, caffe_io, cnn, evaluation
    w = cnn.synthetic_weights(5)
    mean = cnn.synthetic_mean(5)
    layers = {("fc8_20x20" if k == "fc8" else k): [v[0], v[1]] for k, v in w.items()}
    caffe_io.write_caffemodel(str(tmp_path / "weights.caffemodel"), layers)
    caffe_io.write_binaryproto(str(tmp_path / "mean.binaryproto"), mean.reshape(1, 1, 500, 500))
    ds = benchmark.synthetic_dataset("york", str(tmp_path), 3, True)
    evaluation.run_cnn(ds, None, str(tmp_path / "weights.caffemodel"), str(tmp_path / "mean.binaryproto"), gpu=0)
    sphere = np.stack([evaluation._load_pickle(f)['sphere_image'] for f in ds['pickle_files']])
    ref = cnn_torch.forward(w, mean, sphere)
    for f, r in zip(ds['pickle_files'], ref):
        pred = evaluation._load_pickle(f)['cnn_prediction']
        assert pred.shape == (20, 20) and pred.dtype == np.float32
        assert np.abs(pred - r).max() <= 2e-5
