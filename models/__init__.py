"""Import-path shim: the reference's scripts do ``from models.matching import Matching`` and
``from models.line_transformer import LineTransformer``; these modules forward to the MI355X-native
implementation in ``linetr_amd``.  Copy this directory over the reference's ``models/`` files of the same
name (keep its superpoint.py / line_detector.py / utils.py) -- see INTEGRATION.md."""
