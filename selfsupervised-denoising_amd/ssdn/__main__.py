"""`python -m ssdn ...` (ssdn/__main__.py of the reference)."""
import sys
from typing import List

import ssdn
import ssdn.cli


def start_cli(args: List[str] = None):
    ssdn.logging_helper.setup()
    return ssdn.cli.start(args if args is not None else sys.argv[1:])


if __name__ == "__main__":
    start_cli()
