from linetr_amd.nn_matcher import nn_matcher, nn_matcher_distmat  # noqa: F401
