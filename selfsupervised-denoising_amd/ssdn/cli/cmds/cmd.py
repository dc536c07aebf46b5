"""Base class of the CLI commands (cli/cmds/cmd.py:9-45 of the reference)."""
from abc import ABC, abstractmethod
from typing import Dict


class Command(ABC):
    @abstractmethod
    def configure(self, parser):
        """attach the command (and its arguments) to an argparse sub-parser collection"""

    @abstractmethod
    def execute(self, args: Dict):
        """run with the parsed arguments"""

    @abstractmethod
    def cmd(self) -> str:
        """the command word"""
