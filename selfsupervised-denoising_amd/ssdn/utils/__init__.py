from ssdn.utils.utils import *  # noqa: F401,F403
from ssdn.utils.utils import compute_ramped_lrate, Metric, MetricDict, TrackedTime, seconds_to_dhms, separator, cd, list_constants  # noqa: F401
from ssdn.utils.data import (rotate, clip_img, calculate_psnr, mse2psnr, tensor2image, save_tensor_image,  # noqa: F401
                             set_color_channels)
from ssdn.utils import n2v_loss  # noqa: F401
from ssdn.utils import noise  # noqa: F401
from ssdn.utils import n2v_ups  # noqa: F401
