from linetr_amd.line_process import *  # noqa: F401,F403
from linetr_amd.line_process import (change_cv2_T_np, filter_by_length, get_angles, get_dist_matrix, get_line_dist,  # noqa: F401
                                     line_tokenizer, point_on_line, preprocess, remove_borders, sample_descriptors)
