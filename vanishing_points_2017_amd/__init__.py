"""MI355X-native vanishing-point hot path (CNN forward + EM refinement) behind the call surface
of fkluger/vanishing_points_2017's evaluation.py / vp_localisation.py.

All compute runs in libvpk.so (hand-written HIP for gfx950, see csrc/ and include/vpk.h);
PyTorch is used for device memory and streams only.  There is no CPU fallback."""
__version__ = "0.1.0"
