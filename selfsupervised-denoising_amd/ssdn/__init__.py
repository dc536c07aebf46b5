"""ssdn -- self-supervised (blind-spot) image denoising, MI355X-native hot path.

The import name is `ssdn` on purpose: checkpoints written by the reference pickle `ssdn.params.*` /
`ssdn.utils.utils.*` globals.
"""
import ssdn.utils as utils  # noqa: F401
import ssdn.cfg as cfg  # noqa: F401
import ssdn.logging_helper as logging_helper  # noqa: F401
import ssdn.params as params  # noqa: F401
from ssdn.utils.utils import *  # noqa: F401,F403
from ssdn.utils.data import *  # noqa: F401,F403
from ssdn.version import __version__  # noqa: F401
