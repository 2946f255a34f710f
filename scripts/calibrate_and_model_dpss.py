#!/usr/bin/env python
"""Command-line interface of the DPSS calibration + foreground modeling (counterpart of the reference's
scripts/calibrate_and_model_dpss.py): parse the arguments, hand them to the file driver."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from calamity_amd import calibration  # noqa: E402

ap = calibration.dpss_fit_argparser()
args = ap.parse_args()
calibration.read_calibrate_and_model_dpss(**vars(args))
