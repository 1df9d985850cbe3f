from linetr_amd.line_transformer import *  # noqa: F401,F403
from linetr_amd.line_transformer import LineTransformer, get_dist_matrix  # noqa: F401
