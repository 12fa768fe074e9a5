"""Drop-in for the reference's vp_localisation.expectation_maximisation (vp_localisation.py:168-450),
backed by the HIP EM kernels.  Same name, argument order, defaults, in-place normalisation of
``l`` and result keys (including 'distribution', the PDF tuple of the last E-step); `distance_measure` other than
"angle" raises AssertionError as at :203."""
import numpy as np

from . import em as _em


def expectation_maximisation(l, lp, cnn_response, num_iter=100, sphere_image=None, init_vp=None,
                             do_merge=True, do_split=True, do_iterations=True, distance_measure="angle",
                             use_weights=True, wbias=1, num_init_vp=25, split_merge_freq=10,
                             merge_thresh=1e-3, outlier_thresh=1.96 ** 2, final_convergence=5e-3,
                             s_thresh=1e-200, num_min_lines=3, device=0, want_metric=True, return_distribution=True):
    if sphere_image is None:
        raise TypeError("sphere_image is required (the reference dereferences it at vp_localisation.py:113)")
    scene = {"l": l, "lp": lp, "cnn_response": cnn_response, "sphere_image": sphere_image,
             "init_vp": init_vp}
    # 'distribution' is the reference's probability_functions.PDF tuple (:441), as in the reference's result
    res = _em.em_batch([scene], device=device, want_metric=want_metric, want_distribution=return_distribution,
                       num_iter=num_iter,
                       do_merge=do_merge, do_split=do_split, do_iterations=do_iterations,
                       distance_measure=distance_measure, use_weights=use_weights, wbias=wbias,
                       num_init_vp=num_init_vp, split_merge_freq=split_merge_freq,
                       merge_thresh=merge_thresh, outlier_thresh=outlier_thresh,
                       final_convergence=final_convergence, s_thresh=s_thresh,
                       num_min_lines=num_min_lines)[0]
    status = res.pop("status")
    l_norm = res.pop("l")
    if isinstance(l, np.ndarray) and l.dtype == np.float64:
        l[...] = l_norm                       # the reference normalises the caller's array (:185-186)
    if status == 2:                           # np.vstack([]) at vp_localisation.py:165
        raise ValueError("need at least one array to concatenate")
    return res
