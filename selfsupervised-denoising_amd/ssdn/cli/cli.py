"""`ssdn` command line (cli/cli.py:8-44 of the reference): `ssdn train start ...`, `ssdn train resume RUN_DIR`, `ssdn eval ...`."""
import argparse
from typing import List, Optional

from ssdn.version import __version__


def build_parser():
    from ssdn.cli.cmds.eval import EvaluateCommand
    from ssdn.cli.cmds.train import TrainCommand
    parser = argparse.ArgumentParser(prog="ssdn", description="Training and evaluation of blind-spot denoisers (SSDN, Noise2Clean, "
                                     "Noise2Noise, Noise2Void) on AMD MI355X.")
    parser.add_argument("--version", action="version", version="%(prog)s v" + __version__)
    subs = parser.add_subparsers(dest="command", required=True)
    cmds = {}
    for c in (TrainCommand(), EvaluateCommand()):
        c.configure(subs)
        cmds[c.cmd()] = c
    return parser, cmds


def start(argv: Optional[List[str]] = None):
    parser, cmds = build_parser()
    args = vars(parser.parse_args(argv))
    args["PARSER"] = parser
    return cmds[args["command"]].execute(args)
