"""Single-image example with the reference's flags (example.py:11-14): --gpu, --show.

With --source_folder DIR (the reference uses assets/examples, example.py:26) the images of DIR go through the
reference's three calls -- create_data_pickles(cnn_input_size=500, target_size=640), run_cnn, run_em (example.py:36-39)
-- with this package's front end (frontend.py), a random-init CNN unless trained weights are configured (config.py),
and the horizon end points are printed in pixel coordinates like example.py:66-76.  Without it one seeded synthetic
scene (configs[0]: N = 800) goes through raster, CNN and EM."""
import argparse

import numpy as np

from . import calc_horizon, cnn, evaluation, synth


def run_folder(args):
    import os
    from . import config
    os.makedirs(args.destination_folder, exist_ok=True)
    dataset = evaluation.get_data_list(args.source_folder, args.destination_folder, 'default_net', "", "0",
                                       distance_measure='angle', use_weights=True, do_split=True, do_merge=True,
                                       update=True)                                   # example.py:30-34
    evaluation.create_data_pickles(dataset, update=True, cnn_input_size=500, target_size=640)       # :37
    if all(os.path.isfile(f) for f in (config.cnn_weights_path, config.cnn_mean_path)):
        evaluation.run_cnn(dataset, mean_file=config.cnn_mean_path, model_def=config.cnn_config_path,
                           model_weights=config.cnn_weights_path, gpu=args.gpu)          # :38
    else:
        print("no trained weights at %s: random-init AlexNet-500" % config.cnn_weights_path)
        evaluation.run_cnn(dataset, None, None, None, gpu=args.gpu,
                           net=cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), device=args.gpu))
    evaluation.run_em(dataset)                                                        # :39
    for image_file, data_file in zip(dataset['image_files'], dataset['pickle_files']):
        datum = evaluation._load_pickle(data_file)
        res = datum['EM_result']
        height, width = datum['lines']['image_shape'][:2]
        print(image_file, "%d x %d, %d line segments" % (width, height, datum['lines']['line_segments'].shape[0]))
        if res is None or res['vp'] is None:
            print("  no vanishing points")
            continue
        hp1, hp2, _, _, _, _ = calc_horizon.calculate_horizon_and_ortho_vp(res, maxbest=20, theta_vmin=np.pi / 10.)
        scale = max(width, height)
        for hp in (hp1, hp2):                                                          # :66-76
            hp[0] = hp[0] * scale / 2.0 + width / 2.0
            hp[1] = -hp[1] * scale / 2.0 + height / 2.0
        print("  VPs: %d, iterations: %d, horizon: (%.1f, %.1f) - (%.1f, %.1f)" % (
            res['vp'].shape[0], res['iterations'], hp1[0], hp1[1], hp2[0], hp2[1]))
    return dataset


def main(argv=None):
    p = argparse.ArgumentParser(description='')
    p.add_argument('--gpu', default=0, type=int, help='GPU ID to use')
    p.add_argument('--show', dest='show', action='store_true', help='Show results (prints only)')
    p.add_argument('--seed', default=1000, type=int)
    p.add_argument('--lines', default=800, type=int)
    p.add_argument('--source_folder', default=None, help='folder with images (jpg / png / pgm)')
    p.add_argument('--destination_folder', default='/tmp/vp_example_results')
    args = p.parse_args(argv)
    if args.source_folder:
        return run_folder(args)
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
