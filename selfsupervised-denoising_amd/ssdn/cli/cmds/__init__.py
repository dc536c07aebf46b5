from ssdn.cli.cmds.cmd import Command  # noqa: F401
