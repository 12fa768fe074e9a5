"""Record types of the reference's probability_functions.py (:4-5), kept so that result pickles written
by the reference can be read: its EM_result['distribution'] is a ``probability_functions.PDF`` instance
(vp_localisation.py:441), which unpickles only if a class of that name can be found.  The arithmetic of
that module (prior, E-step) lives in the HIP library (csrc/em_device.hpp: prior_setup, estep)."""
from collections import namedtuple

PDFParams = namedtuple('PDFParams', 'means weights sigma')     # probability_functions.py:4
PDF = namedtuple('PDF', 'v lv vl l lvsq angles')               # probability_functions.py:5
