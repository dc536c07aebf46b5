from ssdn.cli.cli import start  # noqa: F401
