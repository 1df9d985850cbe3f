from linetr_amd.line_transformer import (change_cv2_T_np, filter_by_length, get_angles, get_dist_matrix,  # noqa: F401
                                         remove_borders)
