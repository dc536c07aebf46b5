from ssdn.utils.utils import *  # noqa: F401,F403
from ssdn.utils.utils import compute_ramped_lrate, Metric, MetricDict, TrackedTime, seconds_to_dhms, separator, cd  # noqa: F401
from ssdn.utils.data import rotate, clip_img, calculate_psnr, mse2psnr  # noqa: F401
from ssdn.utils import n2v_loss  # noqa: F401
