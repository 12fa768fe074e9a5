"""Single-image example with the reference's flags (example.py:11-14): --gpu, --show.

The reference's example needs LSD on the bundled JPEGs (not vendored); here one seeded synthetic
scene goes through the same three stages -- sphere raster, CNN, EM -- and the horizon end points are
printed in pixel coordinates of a 640-px image like example.py:66-76."""
import argparse

import numpy as np

from . import calc_horizon, cnn, evaluation, synth


def main(argv=None):
    p = argparse.ArgumentParser(description='')
    p.add_argument('--gpu', default=0, type=int, help='GPU ID to use')
    p.add_argument('--show', dest='show', action='store_true', help='Show results (prints only)')
    p.add_argument('--seed', default=1000, type=int)
    p.add_argument('--lines', default=800, type=int)
    args = p.parse_args(argv)
    sc = synth.make_scene(args.seed, args.lines, 3, aspect_h=0.667, raster=None)
    datum = {'lines': {'lines': sc["l"], 'line_segments': sc["lp"], 'image_shape': sc["image_shape"]},
             'sphere_image': evaluation.get_sphere_image(sc["l"], size=500, alpha=0.1)}
    net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), device=args.gpu)
    datum['cnn_prediction'] = cnn.caffe_forward(net, datum['sphere_image'])
    print("CNN response (random-init weights): min %.3f max %.3f" % (datum['cnn_prediction'].min(),
                                                                     datum['cnn_prediction'].max()))
    datum['cnn_prediction'] = sc["cnn_response"]       # stand-in for a trained net's prediction
    datum = evaluation.run_em_single(datum)
    res = datum['EM_result']
    hp1, hp2, _, _, _, _ = calc_horizon.calculate_horizon_and_ortho_vp(res, maxbest=20, theta_vmin=np.pi / 10.)
    height, width = sc["image_shape"]
    for hp in (hp1, hp2):
        hp[0] = hp[0] * 640 / 2.0 + width / 2.0
        hp[1] = -hp[1] * 640 / 2.0 + height / 2.0
    print(hp1)
    print(hp2)
    print("VPs: %d, iterations: %d, inlier lines: %d / %d" % (res['vp'].shape[0], res['iterations'],
                                                           int((res['vp_assoc'] >= 0).sum()), args.lines))


if __name__ == "__main__":
    main()
