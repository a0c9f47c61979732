"""Command-line plumbing shared by the drop-in scripts: every script lists its options as rows
(flag, type, default, help) so that flag names and defaults — the part of the reference's CLIs a caller such as
looper.py depends on — sit in one table per script."""
import argparse


def flag(name, help, type=None, default=None, metavar=None, **more):
    row = dict(help=help, default=default, **more)
    if type is not None:
        row["type"] = type
    if metavar is not None:
        row["metavar"] = metavar
    return name, row


def switch(name, help):
    return name, dict(action="store_true", help=help)


def parse(description, rows, argv=None):
    cli = argparse.ArgumentParser(description=description, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for name, row in rows:
        cli.add_argument(name, **row)
    return cli.parse_args(argv)
