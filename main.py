#!/usr/bin/env python
"""Command-line entry point of the MI355X build: ``python main.py generate_embeddings ...`` (flags: label_anything/cli.py).

Only the hot-path subcommand exists here; the experiment / demo subcommands of the upstream project are out of scope
(DESIGN.md section 1).
"""
import sys


def run(argv=None) -> int:
    import click
    from label_anything import cli
    try:
        cli.main(args=argv, standalone_mode=False)
    except click.ClickException as exc:       # bad flags etc.: print the usage error like click's standalone mode does
        exc.show()
        return exc.exit_code
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
